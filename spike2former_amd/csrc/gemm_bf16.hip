// Spike GEMMs whose activation operand ARRIVES in bf16 (gfx950).
//
// The neuron kernels write spikes -- multiples of 1/D, at most 8 significant bits -- as bf16 (2 bytes / element instead of
// 4), channel-major [batch, K, N] with N contiguous, exactly the layout the fp32 path used.  With the operand already in
// the matrix cores' input type the K loops below contain loads, LDS traffic and MFMAs only:
//   forward  Y[b] = W @ X[b]          X tile copied global -> LDS as it lies in memory ([k][n], 16-byte chunks, no
//                                     conversion, no register transposition); the MFMA B fragments (k-contiguous per
//                                     column) are formed by the LDS transpose read ds_read_b64_tr_b16;
//   dW       dW = sum_b dY[b] X[b]^T  X rows are contraction-contiguous: copied as they lie, read as ds_read_b128.
// W is pre-split into hi + mid + lo bf16 terms (s2f_split_bf16x3), dY is split in the kernel: exact products, fp32
// accumulation, the accuracy of an fp32 GEMM (see gemm.hip).
// Reference call sites: every 1x1 Conv2d / Conv1d / kxk convolution fed by a Q_IFNode (sdtv2.py:121-125, 197-204, 229-235,
// 304-306, 399-405; mmcv_spike/transformer.py:213-236, 758-763; pixel_decoder.py:368-404; SNN_core.py:31-45).
#include "gemm_common.h"
#include <cstdlib>

#pragma clang fp contract(fast)

namespace {

constexpr int BN = 128;
constexpr int BK = 32;
constexpr int LDA = 40;          // A rows in LDS: 32 + 8 bf16 = 80 bytes (conflict-free ds_read_b128 fragments)

// 4 consecutive pixels n .. n+3 of one bf16 plane shifted by the 3x3 tap: ONE 8-byte load at 2-byte alignment; the pixel
// that falls off the image row at its left / right end is zeroed, a row outside the plane reads as zeros.
// Split in three so that the LOAD can be issued a K step ahead with nothing depending on it: the predicate, the (always
// valid) element offset, and the fix-up applied when the tile is staged.  With the select next to the load the compiler
// sinks the load under the predicate and waits for it on the spot (s_waitcnt vmcnt(0) per load).
struct Conv3Pred {
  bool ok, cut_l, cut_r;
};
__device__ __forceinline__ Conv3Pred conv3_pred(int y, int x, int ky, int kx, Conv3 g, bool ok) {
  const int yy = y + ky - 1;
  return Conv3Pred{ok && yy >= 0 && yy < g.H, kx == 0 && x == 0, kx == 2 && x + 4 == g.W};
}
// never addresses outside the plane: at a cut end the aligned neighbour group is loaded and shifted in registers
__device__ __forceinline__ int conv3_off(int n, int ky, int kx, Conv3 g, Conv3Pred p) {
  return p.ok ? n + (ky - 1) * g.W + (kx - 1) + (p.cut_l ? 1 : 0) - (p.cut_r ? 1 : 0) : 0;
}
__device__ __forceinline__ u32x2 conv3_fix(u32x2 v, Conv3Pred p) {
  if (!p.ok) return u32x2{0u, 0u};
  if (p.cut_l) return u32x2{v.x << 16, (v.y << 16) | (v.x >> 16)};          // {0, p0, p1, p2}
  if (p.cut_r) return u32x2{(v.x >> 16) | (v.y << 16), v.y >> 16};          // {p1, p2, p3, 0}
  return v;
}
__device__ __forceinline__ u32x2 conv3_load_bf16(const unsigned short* __restrict__ plane, int n, int y, int x, int ky,
                                                 int kx, Conv3 g, bool ok) {
  const Conv3Pred p = conv3_pred(y, x, ky, kx, g, ok);
  return conv3_fix(*reinterpret_cast<const u32x2*>(plane + conv3_off(n, ky, kx, g, p)), p);
}

// General form of the forward kernel (the mask contraction with the mask_feature convolution folded into it, ops.py):
// a weight per batch element (a_batch_stride elements apart), the contraction running over `K / k_inner` slabs of the
// activation that lie x_outer_stride elements apart (the T time slices of a [T, B, C, HW] map as ONE contraction of length
// T*C), a per-batch row bias, an output scale.  general == 0: the plain layout, every other field unused.
struct GemmEx {
  int general;
  int k_inner;
  int64_t a_batch_stride, x_batch_stride, x_outer_stride, bias_batch_stride;
  float out_scale;
};

template <int CH>
struct Chunk;
template <>
struct Chunk<8> {
  typedef u32x4 type;
};
template <>
struct Chunk<4> {
  typedef u32x2 type;
};

// ---------------------------------------------------------------------------------------------------------------------
// Forward.  Block = WM x 2 wavefronts (x KG split-K groups), wavefront tile 64 x 64 = 2 x 2 MFMA tiles, block tile
// (64 WM) x 128, K step 32 -- the tiling of spike_gemm_kernel (gemm.hip).
// X tile in LDS: [32 k][128 n] bf16, rows of 256 bytes, the 64-byte chunk c of row k stored at chunk c ^ (k & 3): the four
// rows k0 .. k0+3 that the 32 lanes of one transpose read touch then cover all 64 banks once.
// CH = bf16 elements per staged chunk: 8 (16-byte loads, N % 8 == 0) or 4 (8-byte loads: N % 4 == 0, and the implicit
// 3x3 mode whose shifted rows are only 2-byte aligned).
// Wavefronts per SIMD the register allocation must leave room for (HIP's second launch-bound; the LDS tiles allow them):
// without the hint the 128-row tile took 172 registers = two workgroups per CU of four possible.
#ifndef FWD_WM4_WAVES
#define FWD_WM4_WAVES 2
#endif
#ifndef S2F_FWD_OCC
#define FWD_MIN_BLOCKS(WMV, KGV) ((KGV) == 4 ? 2 : (KGV) == 2 ? 3 : (WMV) == 4 ? FWD_WM4_WAVES : 3)      /* wavefronts per SIMD */
#else
#define FWD_MIN_BLOCKS(WMV, KGV) 1
#endif
template <int WM, int TERMS, int CH, bool CONV, int KG, int WNW = 2>
__global__ __launch_bounds__(64 * WNW * WM * KG, FWD_MIN_BLOCKS(WM, KG)) void sgemm_bf16_kernel(const unsigned short* __restrict__ Wsplit,
                                                                   const unsigned short* __restrict__ X,
                                                                   const float* __restrict__ bias, float* __restrict__ Y,
                                                                   int M, int N, int K, int Mpad, int Kpad, int n_tiles,
                                                                   int m_tiles, Conv3 geo, GemmEx ex) {
  constexpr int BM = 64 * WM;
  constexpr int T = 64 * WNW * WM;
  constexpr int NJ = 4 / WNW;                         // 32-column MFMA tiles per wavefront (wave tile 64 x 32 NJ)
  constexpr int AH = 4 / WNW;                         // 16-byte A chunks per thread, term and K step
  constexpr int A_ELEMS = TERMS * BM * LDA;
  constexpr int GROUP_ELEMS = A_ELEMS + BK * BN;
  constexpr int RED_BYTES = (KG - 1) * T * 64 * 4;
  constexpr int LDS_BYTES = KG * GROUP_ELEMS * 2 > RED_BYTES ? KG * GROUP_ELEMS * 2 : RED_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  const int grp = KG > 1 ? threadIdx.x / T : 0;
  const int tid = KG > 1 ? threadIdx.x - grp * T : threadIdx.x;
  unsigned short(*As)[BM][LDA] = reinterpret_cast<unsigned short(*)[BM][LDA]>(smem + (size_t)grp * GROUP_ELEMS * 2);
  unsigned short* Bs = reinterpret_cast<unsigned short*>(smem + (size_t)grp * GROUP_ELEMS * 2 + (size_t)A_ELEMS * 2);

  const int tiles = n_tiles * m_tiles;
  int pid = blockIdx.x;
  if (tiles % 8 == 0) pid = (pid % 8) * (tiles / 8) + pid / 8;          // XCD-aware tile order (see gemm.hip)
  const int mt = pid % m_tiles, nt = pid / m_tiles;
  const int b = blockIdx.y;
  const int m0 = mt * BM, n0 = nt * BN;
  const unsigned short* Xb = X + (ex.general ? (int64_t)b * ex.x_batch_stride : (int64_t)b * (CONV ? K / 9 : K) * N);
  float* Yb = Y + (int64_t)b * M * N;
  Wsplit += (int64_t)b * ex.a_batch_stride;
  if (bias) bias += (int64_t)b * ex.bias_batch_stride;

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNW, wn = wave % WNW;

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t term_stride = (int64_t)Mpad * Kpad;
  constexpr int CPR = BN / CH;                        // chunks per X row
  constexpr int NQ = BK * CPR / T;                    // chunks per thread and K step
  static_assert(NQ >= 1, "tile too small for the thread count");
  static_assert(KG == 1 || WNW == 2, "the split-K groups assume 2 x 2 MFMA tiles per wavefront");
  typedef typename Chunk<CH>::type chunk_t;
  u32x4 areg[TERMS][AH];
  chunk_t breg[NQ];

  const int kloop = (Kpad + KG * BK - 1) / (KG * BK) * (KG * BK);
  auto fetch = [&](int kk) __attribute__((always_inline)) {
    const bool live = KG == 1 || kk < Kpad;
#pragma unroll
    for (int t = 0; t < TERMS; ++t)
#pragma unroll
      for (int h = 0; h < AH; ++h) {
        const int c = tid + h * T;
        areg[t][h] = live ? *reinterpret_cast<const u32x4*>(Wsplit + t * term_stride + (int64_t)(m0 + (c >> 2)) * Kpad + kk +
                                                           (c & 3) * 8)
                          : u32x4{0u, 0u, 0u, 0u};
      }
    const int tap = CONV ? kk / geo.C : 0, ky = tap / 3, kx = tap - 3 * ky;       // uniform over the step (C % 32 == 0)
    const int c0 = CONV ? kk - tap * geo.C : 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = tid + q * T;
      const int kr = p / CPR, nc = p % CPR;
      const int n = n0 + nc * CH;
      const bool ok = live && kk + kr < K && n < N;          // N % CH == 0: a chunk is valid as a whole
      if constexpr (CONV) {
        // raw load; the border fix-up happens when the tile is staged (conv3_fix), one K step later
        const int py = n / geo.W, px = n - py * geo.W;
        const Conv3Pred pr = conv3_pred(py, px, ky, kx, geo, ok);
        breg[q] = *reinterpret_cast<const chunk_t*>(Xb + (int64_t)(ok ? c0 + kr : 0) * N + conv3_off(n, ky, kx, geo, pr));
      } else {
        // general form: row r of the contraction lies in slab r / k_inner (slabs x_outer_stride apart; k_inner % 32 == 0, so
        // a K step never straddles two slabs)
        // bare load from an always-valid address (row / column 0 when out of range); zeroed when the tile is staged
        const int krc = ok ? kr : -kk;                      // kk + krc == 0
        const int64_t rowoff = ex.general ? (int64_t)(kk / ex.k_inner) * ex.x_outer_stride + (int64_t)(kk % ex.k_inner + kr) * N
                                          : (int64_t)(kk + krc) * N;
        breg[q] = *reinterpret_cast<const chunk_t*>(Xb + (ex.general && !ok ? 0 : rowoff) + (ok ? n : 0));
      }
    }
  };

  // per-lane part of the transpose-read addresses (bf16 elements): row 8 (lane >> 5) + kl, kl = (lane & 15) >> 2; inside the
  // 32-column chunk the lane addresses columns 16 ((lane >> 4) & 1) + 4 (lane & 3)
  const int kl = (lane & 15) >> 2;
  int boff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    boff[j] = (8 * (lane >> 5) + kl) * BN + (((wn * NJ + j) ^ kl) << 5) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  fetch(grp * BK);
  for (int k0 = grp * BK; k0 < kloop; k0 += KG * BK) {
#pragma unroll
    for (int t = 0; t < TERMS; ++t)
#pragma unroll
      for (int h = 0; h < AH; ++h) {
        const int c = tid + h * T;
        *reinterpret_cast<u32x4*>(&As[t][c >> 2][(c & 3) * 8]) = areg[t][h];
      }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = tid + q * T;
      const int kr = p / CPR, nc = p % CPR;
      constexpr int CPC = 32 / CH;                        // chunks per 64-byte swizzle unit
      chunk_t bv = breg[q];
      if constexpr (!CONV) {
        const bool ok = (KG == 1 || k0 < Kpad) && k0 + kr < K && n0 + nc * CH < N;
        if (!ok) {
#pragma unroll
          for (int e = 0; e < CH / 2; ++e) bv[e] = 0u;
        }
      }
      if constexpr (CONV) {
        const int tap = k0 / geo.C, ky = tap / 3, kx = tap - 3 * ky;
        const int n = n0 + nc * CH, py = n / geo.W, px = n - py * geo.W;
        const bool ok = (KG == 1 || k0 < Kpad) && k0 + kr < K && n < N;
        bv = conv3_fix(bv, conv3_pred(py, px, ky, kx, geo, ok));
      }
      *reinterpret_cast<chunk_t*>(Bs + kr * BN + ((((nc / CPC) ^ (kr & 3))) << 5) + (nc % CPC) * CH) = bv;
    }
    __syncthreads();
    if (k0 + KG * BK < kloop) fetch(k0 + KG * BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kof = ks * 16 + 8 * (lane >> 5);
      bf16x8 bfrag[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        union {
          bf16x8 v;
          s16x4 h[2];
        } u;
        u.h[0] = s2f_lds_tr16(Bs + boff[j] + (ks * 16) * BN);
        u.h[1] = s2f_lds_tr16(Bs + boff[j] + (ks * 16 + 4) * BN);
        bfrag[j] = u.v;
      }
#pragma unroll
      for (int t = 0; t < TERMS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(&As[t][wm * 64 + i * 32 + (lane & 31)][kof]);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  if (KG > 1) {
    float* red = reinterpret_cast<float*>(smem);
    if (grp > 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(((grp - 1) * 4 + i * 2 + j) * 16 + r) * T + tid] = acc[i][j][r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < KG; ++g)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += red[(((g - 1) * 4 + i * 2 + j) * 16 + r) * T + tid];
  }
  // epilogue: C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wn * (NJ * 32) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float v = acc[i][j][r];
          if (bias) v += bias[row];
          Yb[(int64_t)row * N + col] = ex.general ? v * ex.out_scale : v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient with a bf16 activation:  dW[m][k] = sum_b sum_l dY[b][m][l] X[b][k][l].  The structure of
// spike_gemm_dw_kernel (gemm.hip: 4 waves, output tile TM x 128, split-K over B*L with fp32 atomics, dY split hi+mid+lo
// while it is staged); the X tile is copied as it lies (8-byte chunks of 4 contraction elements).
// ASPLIT: dY arrives PRE-SPLIT as three bf16 planes hi | mid | lo (`plane` elements apart, written by s2f_bn_act_bwd_split): the
// tile is copied as it lies, like X -- no conversion arithmetic in the loop.
template <int BKV, bool CONV, int TM, bool ASPLIT = false>
__device__ __forceinline__ void dw_tile_body(const float* __restrict__ dY, const unsigned short* __restrict__ X,
                                             float* __restrict__ dW, int B, int M, int K, int L, int steps_per_split,
                                             int k_tiles, Conv3 geo, int log_w, int tile, int split, int64_t plane = 0) {
  constexpr int LD = BKV + 8;
  constexpr int QPR = BKV / 4;                 // 4-element chunks per row
  constexpr int NH = 128 * QPR / 256;          // X chunks per thread
  constexpr int NHA = TM * QPR / 256;          // dY chunks per thread
  constexpr int WMW = TM == 128 ? 2 : 1, WNW = 4 / WMW;
  constexpr int MI = TM / WMW / 32, NJ = 128 / WNW / 32;
  static_assert(NHA >= 1 && MI >= 1 && NJ >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) unsigned short As[3][TM][LD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[128][LD];
  const int m0 = (tile / k_tiles) * TM, k0 = (tile % k_tiles) * 128;
  const int lsteps = (L + BKV - 1) / BKV;
  const int total_steps = B * lsteps;
  const int s_begin = split * steps_per_split;
  const int s_end = min(total_steps, s_begin + steps_per_split);
  if (s_begin >= s_end) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNW, wn = wave % WNW;

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[ASPLIT ? 1 : NHA];
  u32x2 aplane[ASPLIT ? 3 : 1][NHA];
  u32x2 breg[NH];
  int crow[NH], ctap[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int k = min(k0 + (tid + h * 256) / QPR, K - 1);
    ctap[h] = CONV ? k / geo.C : 0;
    crow[h] = CONV ? k - ctap[h] * geo.C : 0;
  }
  // Per-lane element offsets inside the (batch element, K step) slab, fixed for the whole kernel; the slab origin is a
  // wave-uniform pointer, so a load is "scalar base + 32-bit lane offset" with no per-step address arithmetic on the VALU
  // (the flat 64-bit form cost 32 v_mul_lo_u32 + 32 64-bit multiply-adds per wavefront and step -- about the MFMA time).
  unsigned int oa[NHA], ox[NH];                 // row part of the offset (elements)
  int lq[NH];
  bool rok_a[NHA], rok_x[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int c = tid + h * 256;
    const int row = c / QPR;
    lq[h] = (c % QPR) * 4;
    if (h < NHA) {
      rok_a[h < NHA ? h : 0] = m0 + row < M;
      oa[h < NHA ? h : 0] = (unsigned int)min(row, M - 1 - m0) * (unsigned int)L;
    }
    rok_x[h] = k0 + row < K;
    ox[h] = (unsigned int)min(row, K - 1 - k0) * (unsigned int)L;
  }
  auto fetch = [&](int step, f32x4 (&a)[ASPLIT ? 1 : NHA], u32x2 (&bq)[NH]) {
    const int b = step / lsteps, l0 = (step - b * lsteps) * BKV;
    const float* pa = dY + ((int64_t)b * M + m0) * L;                            // wave-uniform
    const unsigned short* ps = reinterpret_cast<const unsigned short*>(dY) + ((int64_t)b * M + m0) * L;
    const unsigned short* px = X + ((int64_t)b * K + k0) * L;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      // Unconditional loads from clamped addresses, zeroed by a select: a load under a branch makes the number of loads in
      // flight unknown to the compiler, which then waits with s_waitcnt vmcnt(0) -- i.e. also for the NEWEST prefetch.
      const int l = l0 + lq[h];
      const bool lok = l < L;
      const unsigned int col = (unsigned int)min(l, L - 4);          // L % 4 == 0, L >= 4
      // raw loads only: the out-of-range lanes are zeroed in stage(), one K step later -- with the select next to the load
      // the compiler sinks the load under the predicate and waits for it on the spot
      if constexpr (ASPLIT) {
        if (h < NHA) {
#pragma unroll
          for (int t = 0; t < 3; ++t)
            aplane[t][h < NHA ? h : 0] = *reinterpret_cast<const u32x2*>(ps + t * plane + (oa[h < NHA ? h : 0] + col));
        }
      } else {
        if (h < NHA) a[h < NHA ? h : 0] = *reinterpret_cast<const f32x4*>(pa + (oa[h < NHA ? h : 0] + col));
      }
      if constexpr (CONV) {
        const Conv3Pred pr = conv3_pred(conv3_row(l, geo.W, log_w), conv3_col(l, geo.W, log_w), ctap[h] / 3, ctap[h] % 3, geo, lok && rok_x[h]);
        bq[h] = *reinterpret_cast<const u32x2*>(X + ((int64_t)b * geo.C + crow[h]) * L + conv3_off(l, ctap[h] / 3, ctap[h] % 3, geo, pr));
      } else {
        bq[h] = *reinterpret_cast<const u32x2*>(px + (ox[h] + col));
      }
    }
  };
  auto stage = [&](int step, const f32x4 (&a)[ASPLIT ? 1 : NHA], const u32x2 (&bq)[NH]) __attribute__((always_inline)) {
    const int l0s = (step - (step / lsteps) * lsteps) * BKV;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int c = tid + h * 256;
      const int row = c / QPR, col = (c % QPR) * 4;
      const bool lok = l0s + lq[h] < L;
      if constexpr (ASPLIT) {
        if (h < NHA) {
          const bool ok = lok && rok_a[h < NHA ? h : 0];
#pragma unroll
          for (int t = 0; t < 3; ++t)
            *reinterpret_cast<u32x2*>(&As[t][row][col]) = ok ? aplane[t][h < NHA ? h : 0] : u32x2{0u, 0u};
        }
      } else if (h < NHA) {
        const f32x4 av = (lok && rok_a[h < NHA ? h : 0]) ? a[h < NHA ? h : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned int h0, m0_, l0_, h1, m1, l1;
        s2f_split3x2(av.x, av.y, h0, m0_, l0_);
        s2f_split3x2(av.z, av.w, h1, m1, l1);
        *reinterpret_cast<u32x2*>(&As[0][row][col]) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(&As[1][row][col]) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(&As[2][row][col]) = u32x2{l0_, l1};
      }
      if constexpr (CONV) {
        const int l = l0s + lq[h];
        *reinterpret_cast<u32x2*>(&Bs[row][col]) =
            conv3_fix(bq[h], conv3_pred(conv3_row(l, geo.W, log_w), conv3_col(l, geo.W, log_w), ctap[h] / 3, ctap[h] % 3, geo, lok && rok_x[h]));
      } else {
        *reinterpret_cast<u32x2*>(&Bs[row][col]) = (lok && rok_x[h]) ? bq[h] : u32x2{0u, 0u};
      }
    }
  };
  auto compute = [&]() __attribute__((always_inline)) {
    // (Requesting the fragments of sub-step ks + 1 before the MFMAs of sub-step ks -- two register sets, s_waitcnt lgkmcnt(n)
    // counting down instead of lgkmcnt(0) before every pair of MFMAs -- measured no gain: 184.8 vs 180.6 us.)
#pragma unroll
    for (int ks = 0; ks < BKV / 16; ++ks) {
      const int kof = ks * 16 + 8 * (lane >> 5);
      bf16x8 bfrag[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        bfrag[j] = *reinterpret_cast<const bf16x8*>(&Bs[wn * (NJ * 32) + j * 32 + (lane & 31)][kof]);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(&As[t][wm * (MI * 32) + i * 32 + (lane & 31)][kof]);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[j], acc[i][j], 0, 0, 0);
        }
    }
  };
  // (Two register sets fetched TWO steps ahead: no gain in the probe or in the step -- 45.76/45.81 vs 45.83/45.82 ms.)
  fetch(s_begin, areg, breg);
  for (int step = s_begin; step < s_end; ++step) {
    stage(step, areg, breg);
    __syncthreads();
    if (step + 1 < s_end) fetch(step + 1, areg, breg);
    compute();
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = k0 + wn * (NJ * 32) + j * 32 + (lane & 31);
      // CONV, plane < 0: dW in the WEIGHT's layout [M][C][3][3] -- column (tap, c) of the tap-major product goes to c * 9 + tap
      int colw = col;
      if (CONV && plane < 0) {
        const int tap = col / geo.C;
        colw = (col - tap * geo.C) * 9 + tap;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < K) atomicAdd(dW + (int64_t)row * K + colw, acc[i][j][r]);
      }
    }
}

// Workgroup f of `total` runs on XCD f % 8 (round-robin dispatch); this gives XCD j a CONTIGUOUS range of logical ids.  The
// weight gradient orders its workgroups tile-fastest inside a contraction split, so the workgroups of one XCD are the tiles
// of the same few splits: they stream the same dY / X slices at the same time and share them through that XCD's L2.  With
// the plain order the ~36 tiles of a split were spread over all eight L2s and every tile re-fetched its operands (885 MB of
// L2 misses for 142 MB of operands on [512x1152] over [8x4096]: the kernel ran at that traffic, not at its MFMA rate).
__device__ __forceinline__ int xcd_contiguous(int f, int total) {
  const int chunk = total >> 3, rem = total & 7;
  const int xcd = f & 7, idx = f >> 3;
  return xcd * chunk + min(xcd, rem) + idx;
}

// Workgroups per CU the register allocation must leave room for: the 64-row tiles sat at 172 registers, 4 above the limit for
// three wavefronts per SIMD (the LDS tile allows three workgroups).
#ifndef S2F_DW_OCC
#define DW_MIN_BLOCKS(TMV, BKVV) ((TMV) <= 64 ? 3 : 2)      /* wavefronts per SIMD (= workgroups per CU: 4 wavefronts each) */
#else
#define DW_MIN_BLOCKS(TMV, BKVV) 1
#endif

template <int BKV, bool CONV, int TM, bool ASPLIT>
__global__ __launch_bounds__(256, DW_MIN_BLOCKS(TM, BKV)) void sgemm_dw_bf16_kernel(const float* __restrict__ dY,
                                                            const unsigned short* __restrict__ X, float* __restrict__ dW,
                                                            int B, int M, int K, int L, int steps_per_split, int k_tiles,
                                                            Conv3 geo, int log_w, int64_t plane) {
  const int id = xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  dw_tile_body<BKV, CONV, TM, ASPLIT>(dY, X, dW, B, M, K, L, steps_per_split, k_tiles, geo, log_w, id % (int)gridDim.x,
                                      id / (int)gridDim.x, plane);
}

// MANY weight gradients in ONE launch.  The 32x32- and 64x64-stage layers (and the decoder's 100-token layers) each owe a
// weight gradient of 1-4 GFLOP whose contraction is only B*L = 8 192 .. 32 768 long: as separate launches (~180 per step)
// each one fills the chip only by splitting its contraction 32-64 ways, i.e. 2-4 steps of work per workgroup in front of a
// 32 KiB atomic tile, 18-35 us apiece.  Nothing downstream waits for a weight gradient until the gradients are packed, so
// the host defers them (ops.DEFER_DW) and hands the whole list over at the end of the backward pass: one grid over all
// (job, tile, split) triples, the job table in the kernel arguments (baked into a captured hipGraph like any other
// argument), every workgroup long enough to amortise its atomics.
constexpr int kMaxJobs = 56;
struct DwJob {
  const float* dY;
  const unsigned short* X;
  float* dW;
  int B, M, K, L;
  int first_block, steps_per_split, k_tiles, tiles;
  int64_t plane;                       // > 0: dY is three bf16 planes this many elements apart
};
struct DwJobTable {
  int njobs;
  DwJob job[kMaxJobs];
};

template <int BKV, int TM, bool ASPLIT>
__global__ __launch_bounds__(256, DW_MIN_BLOCKS(TM, BKV)) void sgemm_dw_grouped_kernel(const DwJobTable tab) {
  const int id = xcd_contiguous(blockIdx.x, gridDim.x);
  int lo = 0, hi = tab.njobs - 1;                         // last job whose first block <= id (wave-uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab.job[mid].first_block <= id) lo = mid; else hi = mid - 1;
  }
  const DwJob& j = tab.job[lo];
  const int local = id - j.first_block;
  dw_tile_body<BKV, false, TM, ASPLIT>(j.dY, j.X, j.dW, j.B, j.M, j.K, j.L, j.steps_per_split, j.k_tiles, Conv3{0, 0, 0}, 0,
                                       local % j.tiles, local / j.tiles, j.plane);
}

// (Two larger-tile forms of this kernel were built and measured, then removed -- tools/micro/gemm_dw_probe.hip, us on
// [512x1152] over [8x4096] against 171 for the kernel above: a PRODUCER / CONSUMER split of a 512-thread workgroup over
// double-buffered LDS (wavefronts 4-7 prefetch three steps ahead, split and stage tile s+1 while wavefronts 0-3 multiply tile s
// on 64 x 128 wavefront tiles, one barrier per step): 247; its halves alone: consumers 159, producers 170 -- one wavefront per
// SIMD cannot hide the LDS round trip in front of its MFMAs, nor the staging chain; and a 256 x 256 tile with all eight
// wavefronts staging and multiplying (half the LDS / vL1D / VALU traffic per MFMA): 256.  Cache-resident operands (every step
// re-reading the first slab) leave the time unchanged, as does the XCD-contiguous order below: the loop is bound inside the CU.)
// fp32 -> bf16 of an exactly representable tensor (spikes handed over by a caller that still holds them in fp32):
// truncation == rounding for such values; 4 elements per thread.
__global__ void to_bf16_exact_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
    *reinterpret_cast<u32x2*>(y + 4 * i) = u32x2{s2f_pack2(f32x2{v.x, v.y}), s2f_pack2(f32x2{v.z, v.w})};
  }
}

int fwd_launch(const char* who, const uint16_t* w_split, const uint16_t* X, const float* bias, float* Y, int batch, int M, int N,
               int K, int Mpad, int Kpad, int terms, bool conv, Conv3 geo, void* stream, GemmEx ex = GemmEx{0, 1, 0, 0, 0, 0, 1.f}) {
  S2F_REQUIRE(w_split && X && Y, S2F_EINVAL, "%s: null pointer", who);
  S2F_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && terms >= 1 && terms <= 3, S2F_EINVAL, "%s: bad sizes", who);
  S2F_REQUIRE((N & 3) == 0, S2F_EINVAL, "%s: N=%d must be a multiple of 4", who, N);
  S2F_REQUIRE(Kpad >= K && Kpad % 32 == 0 && Mpad >= M && Mpad % 64 == 0, S2F_EINVAL, "%s: bad padding", who);
  S2F_REQUIRE(s2f_aligned16(w_split) && s2f_aligned16(X) && s2f_aligned16(Y), S2F_EALIGN,
              "%s: pointers must be 16-byte aligned", who);
  S2F_REQUIRE(batch < 65536, S2F_EINVAL, "%s: batch too large", who);
  hipStream_t s = (hipStream_t)stream;
  const int n_tiles = (N + BN - 1) / BN;
  static const int s_min_blocks = getenv("S2F_GEMM_MINBLOCKS") ? atoi(getenv("S2F_GEMM_MINBLOCKS")) : 512;   // probe switch
  int wm = 1;
  for (int cand = 4; cand >= 1; cand >>= 1) {          // tile choice as in gemm.hip
    if (Mpad % (64 * cand) != 0) continue;
    if (cand > 1 && M <= 32 * cand) continue;
    const int64_t blocks = (int64_t)n_tiles * (Mpad / (64 * cand)) * batch;
    // one workgroup per CU is enough for the larger tile: measured (tools/micro/gemm_fwd_probe.hip) [512x512]@[8x512x1024]
    // 28.7 us as 512 tiles of 64 rows, 23.9 us as 256 tiles of 128 rows; [512x1536] 68.5 -> 55.9 us
    if (blocks >= s_min_blocks || cand == 1) {
      wm = cand;
      break;
    }
  }
  static const char* force_wm = getenv("S2F_GEMM_WM");           // probe switch
  if (force_wm && atoi(force_wm) > 0 && Mpad % (64 * atoi(force_wm)) == 0) wm = atoi(force_wm);
  const int m_tiles = Mpad / (64 * wm);
  const dim3 grid(n_tiles * m_tiles, batch);
  const bool thin = wm == 1 && (int64_t)n_tiles * m_tiles * batch < 512;
  // thin launches (fewer than 512 workgroups of 2 wavefronts) are one dependent load -> LDS -> MFMA chain per K step: the
  // contraction is split over 2 or 4 groups of wavefronts inside the workgroup (summed through LDS in a fixed order)
  const int kg = !thin ? 1 : (Kpad >= 16 * BK ? 4 : (Kpad >= 4 * BK ? 2 : 1));
  const bool wide = !conv && (N & 7) == 0;              // 16-byte chunks
  // (A 64 x 128 wavefront tile -- WNW = 1, half the wavefronts, 37 % less LDS read traffic per MFMA -- measured SLOWER:
  // [512x1152]@[8x1152x4096] 119 -> 149 us, [256x256]@[8x256x16384] 67 -> 89 us (tools/micro/gemm_fwd_probe.hip): the loop
  // is bound by the latency of its LDS reads and barriers at two wavefronts per SIMD, not by LDS bandwidth.)
#define S2F_GO(WMV, TV, CHV, CV, KGV)                                                                                    \
  S2F_LAUNCH(true, true, (sgemm_bf16_kernel<WMV, TV, CHV, CV, KGV>), grid, dim3(128 * WMV * KGV), 0, s, w_split, X, bias, Y, \
             M, N, K, Mpad, Kpad, n_tiles, m_tiles, geo, ex)
#define S2F_T(WMV, CHV, CV, KGV)                    \
  if (terms == 3) S2F_GO(WMV, 3, CHV, CV, KGV);      \
  else if (terms == 2) S2F_GO(WMV, 2, CHV, CV, KGV); \
  else S2F_GO(WMV, 1, CHV, CV, KGV)
#define S2F_W(CHV, CV)        \
  if (wm == 4) {              \
    S2F_T(4, CHV, CV, 1);     \
  } else if (wm == 2) {       \
    S2F_T(2, CHV, CV, 1);     \
  } else if (kg == 4) {       \
    S2F_T(1, CHV, CV, 4);     \
  } else if (kg == 2) {       \
    S2F_T(1, CHV, CV, 2);     \
  } else {                    \
    S2F_T(1, CHV, CV, 1);     \
  }
  if (conv) {
    S2F_W(4, true)
  } else if (wide) {
    S2F_W(8, false)
  } else {
    S2F_W(4, false)
  }
#undef S2F_W
#undef S2F_T
#undef S2F_GO
  return s2f_check_launch(who);
}

int dw_launch(const float* dY, const uint16_t* X, float* dW, int batch, int M, int K, int L, int accumulate, bool conv,
              Conv3 geo, int log_w, void* stream, int64_t plane = 0) {
  S2F_REQUIRE(dY && X && dW, S2F_EINVAL, "s2f_spike_gemm_dw_bf16: null pointer");
  S2F_REQUIRE(batch > 0 && M > 0 && K > 0 && L > 0 && (L & 3) == 0, S2F_EINVAL,
              "s2f_spike_gemm_dw_bf16: bad sizes (L=%d must be a positive multiple of 4)", L);
  S2F_REQUIRE(s2f_aligned16(dY) && (reinterpret_cast<uintptr_t>(X) & 7u) == 0, S2F_EALIGN,
              "s2f_spike_gemm_dw_bf16: dY must be 16-byte, X 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (!(accumulate & 1) && s2f_zero_async(dW, sizeof(float) * (size_t)M * K, s) != S2F_OK)
    return s2f_check_launch("s2f_spike_gemm_dw_bf16 memset");
  if (conv && (accumulate & 2)) plane = -1;          // the implicit 3x3 form writing the weight's own layout (see the kernel's epilogue)
  // tile / step / split choice: the measured model of gemm.hip's spike_dw_launch
  int tm = M <= 32 ? 32 : M <= 64 ? 64 : 128;
  if (tm == 128 && (int64_t)batch * L <= 16384 && (int64_t)M * K > 65536) tm = 64;
  const int m_tiles = (M + tm - 1) / tm, k_tiles = (K + 127) / 128;
  const int bkv = (!conv && (L % 64 == 0 || L >= 512)) ? 64 : 32;
  const int total_steps = batch * ((L + bkv - 1) / bkv);
  const int tiles = m_tiles * k_tiles;
  const double t_step = (bkv == 64 ? 2.3 : 1.2) * (tm == 32 ? 0.6 : tm == 64 ? 0.75 : 1.0),
               t_mb = 0.6 * (double)M * K * 4.0 / 1e6;
  int splits = 1;
  double best = 1e30;
  for (int cand = 1; cand <= total_steps && cand <= 65535; cand *= 2) {
    const int steps = (total_steps + cand - 1) / cand;
    const int64_t rounds = ((int64_t)tiles * cand + 511) / 512;
    const double cost = (double)steps * t_step * (double)rounds + (double)cand * t_mb;
    if (cost < best) {
      best = cost;
      splits = cand;
    }
  }
  if (splits > total_steps) splits = total_steps;
  if (splits > 65535) splits = 65535;
  const int steps_per_split = (total_steps + splits - 1) / splits;
  splits = (total_steps + steps_per_split - 1) / steps_per_split;
#define S2F_DW1(BKV, CV, TMV)                                                                                            \
  do {                                                                                                                   \
    if (plane > 0)                                                                                                       \
      S2F_LAUNCH(true, true, (sgemm_dw_bf16_kernel<BKV, CV, TMV, true>), dim3(m_tiles * k_tiles, splits), dim3(256), 0, s, dY, \
                 X, dW, batch, M, K, L, steps_per_split, k_tiles, geo, log_w, plane);                                     \
    else                                                                                                                 \
      S2F_LAUNCH(true, true, (sgemm_dw_bf16_kernel<BKV, CV, TMV, false>), dim3(m_tiles * k_tiles, splits), dim3(256), 0, s, dY, \
                 X, dW, batch, M, K, L, steps_per_split, k_tiles, geo, log_w, plane);                                     \
  } while (0)
#define S2F_DW(BKV, CV)             \
  do {                              \
    if (tm == 32)                   \
      S2F_DW1(BKV, CV, 32);         \
    else if (tm == 64)              \
      S2F_DW1(BKV, CV, 64);         \
    else                            \
      S2F_DW1(BKV, CV, 128);        \
  } while (0)
  if (conv)
    S2F_DW(32, true);
  else if (bkv == 64)
    S2F_DW(64, false);
  else
    S2F_DW(32, false);
#undef S2F_DW
#undef S2F_DW1
  return s2f_check_launch("s2f_spike_gemm_dw_bf16");
}

}  // namespace

extern "C" int s2f_to_bf16_exact(const float* x, uint16_t* y, int64_t n, void* stream) {
  if (n == 0) return S2F_OK;
  S2F_REQUIRE(x && y && n > 0 && (n & 3) == 0, S2F_EINVAL, "s2f_to_bf16_exact: null pointer or n %% 4 != 0");
  S2F_REQUIRE(s2f_aligned16(x) && (reinterpret_cast<uintptr_t>(y) & 7u) == 0, S2F_EALIGN, "s2f_to_bf16_exact: alignment");
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(to_bf16_exact_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n / 4);
  return s2f_check_launch("s2f_to_bf16_exact");
}

extern "C" int s2f_spike_gemm_fwd_bf16(const uint16_t* w_split, const uint16_t* X, const float* bias, float* Y, int batch,
                                       int M, int N, int K, int Mpad, int Kpad, int terms, void* stream) {
  return fwd_launch("s2f_spike_gemm_fwd_bf16", w_split, X, bias, Y, batch, M, N, K, Mpad, Kpad, terms, false, Conv3{0, 0, 0},
                    stream);
}

extern "C" int s2f_spike_conv3x3_fwd_bf16(const uint16_t* w_split, const uint16_t* X, const float* bias, float* Y, int batch,
                                          int M, int C, int H, int W, int Mpad, int Kpad, int terms, void* stream) {
  S2F_REQUIRE(C > 0 && C % 32 == 0 && H > 0 && W > 0 && (W & 3) == 0, S2F_EINVAL,
              "s2f_spike_conv3x3_fwd_bf16: need C %% 32 == 0 and W %% 4 == 0 (C=%d, W=%d)", C, W);
  S2F_REQUIRE((int64_t)C * 9 < (1 << 30) && (int64_t)H * W < (1 << 30), S2F_EINVAL, "s2f_spike_conv3x3_fwd_bf16: too large");
  return fwd_launch("s2f_spike_conv3x3_fwd_bf16", w_split, X, bias, Y, batch, M, H * W, C * 9, Mpad, Kpad, terms, true,
                    Conv3{H, W, C}, stream);
}

extern "C" int s2f_spike_gemm_dw_bf16(const float* dY, const uint16_t* X, float* dW, int batch, int M, int K, int L,
                                      int accumulate, void* stream) {
  return dw_launch(dY, X, dW, batch, M, K, L, accumulate, false, Conv3{0, 0, 0}, 0, stream);
}

extern "C" int s2f_spike_gemm_dw_bf16_split(const uint16_t* dY_split, int64_t plane_stride, const uint16_t* X, float* dW, int batch,
                                            int M, int K, int L, int accumulate, void* stream) {
  S2F_REQUIRE(plane_stride >= (int64_t)batch * M * L && (plane_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(dY_split) & 15u) == 0,
              S2F_EINVAL, "s2f_spike_gemm_dw_bf16_split: plane stride must cover a plane and keep 8-byte alignment");
  return dw_launch(reinterpret_cast<const float*>(dY_split), X, dW, batch, M, K, L, accumulate, false, Conv3{0, 0, 0}, 0, stream,
                   plane_stride);
}

extern "C" int s2f_spike_conv3x3_dw_bf16(const float* dY, const uint16_t* X, float* dW, int batch, int M, int C, int H, int W,
                                         int accumulate, void* stream) {
  S2F_REQUIRE(C > 0 && C % 32 == 0 && H > 0 && W >= 4 && (W & 3) == 0, S2F_EINVAL,
              "s2f_spike_conv3x3_dw_bf16: need C %% 32 == 0 and W %% 4 == 0 (C=%d, W=%d)", C, W);
  int log_w = -1;                                // any width: the kernel divides (C5's maps are 672 / 336 / 168 / 84 wide)
  if ((W & (W - 1)) == 0)
    for (log_w = 0; (1 << log_w) < W;) ++log_w;
  return dw_launch(dY, X, dW, batch, M, C * 9, H * W, accumulate, true, Conv3{H, W, C}, log_w, stream);
}

static int dw_grouped_launch(const int64_t* jobs, int njobs, int bkv, void* stream, bool split);

extern "C" int s2f_spike_gemm_dw_grouped(const int64_t* jobs, int njobs, int bkv, void* stream) {
  return dw_grouped_launch(jobs, njobs, bkv, stream, false);
}

extern "C" int s2f_spike_gemm_dw_grouped_split(const int64_t* jobs, int njobs, int bkv, void* stream) {
  return dw_grouped_launch(jobs, njobs, bkv, stream, true);
}

static int dw_grouped_launch(const int64_t* jobs, int njobs, int bkv, void* stream, bool split) {
  // jobs (HOST array): njobs x {dY, X, dW (pointers), batch, M, K, L}; every dW is accumulated into (accumulate semantics);
  // split: dY points at three bf16 planes hi | mid | lo, batch * M * L elements apart
  S2F_REQUIRE(jobs && njobs > 0 && njobs <= kMaxJobs && (bkv == 32 || bkv == 64), S2F_EINVAL,
              "s2f_spike_gemm_dw_grouped: 1 .. %d jobs, bkv 32 or 64", kMaxJobs);
  DwJobTable tab;
  tab.njobs = njobs;
  int64_t work = 0;
  for (int i = 0; i < njobs; ++i) {
    const int64_t* r = jobs + 7 * i;
    DwJob& j = tab.job[i];
    j.dY = reinterpret_cast<const float*>(r[0]);
    j.X = reinterpret_cast<const unsigned short*>(r[1]);
    j.dW = reinterpret_cast<float*>(r[2]);
    j.B = (int)r[3], j.M = (int)r[4], j.K = (int)r[5], j.L = (int)r[6];
    S2F_REQUIRE(j.dY && j.X && j.dW && j.B > 0 && j.M > 0 && j.K > 0 && j.L > 0 && (j.L & 3) == 0, S2F_EINVAL,
                "s2f_spike_gemm_dw_grouped: bad job %d", i);
    S2F_REQUIRE(s2f_aligned16(j.dY) && (reinterpret_cast<uintptr_t>(j.X) & 7u) == 0, S2F_EALIGN,
                "s2f_spike_gemm_dw_grouped: job %d misaligned", i);
    j.k_tiles = (j.K + 127) / 128;
    j.tiles = ((j.M + 63) / 64) * j.k_tiles;
    j.plane = split ? (int64_t)j.B * j.M * j.L : 0;
    work += (int64_t)j.tiles * j.B * ((j.L + bkv - 1) / bkv);
  }
  // one contraction length per workgroup for the whole launch: ~1024 workgroups, at least 4 steps each
  int sps = (int)((work + 1023) / 1024);
  if (sps < 4) sps = 4;
  int64_t first = 0;
  for (int i = 0; i < njobs; ++i) {
    DwJob& j = tab.job[i];
    const int steps = j.B * ((j.L + bkv - 1) / bkv);
    j.steps_per_split = sps < steps ? sps : steps;
    const int splits = (steps + j.steps_per_split - 1) / j.steps_per_split;
    j.first_block = (int)first;
    first += (int64_t)j.tiles * splits;
  }
  S2F_REQUIRE(first < (1ll << 31), S2F_EINVAL, "s2f_spike_gemm_dw_grouped: grid too large");
  hipStream_t s = (hipStream_t)stream;
  if (bkv == 64 && split)
    S2F_LAUNCH(true, true, (sgemm_dw_grouped_kernel<64, 64, true>), dim3((unsigned)first), dim3(256), 0, s, tab);
  else if (bkv == 64)
    S2F_LAUNCH(true, true, (sgemm_dw_grouped_kernel<64, 64, false>), dim3((unsigned)first), dim3(256), 0, s, tab);
  else if (split)
    S2F_LAUNCH(true, true, (sgemm_dw_grouped_kernel<32, 64, true>), dim3((unsigned)first), dim3(256), 0, s, tab);
  else
    S2F_LAUNCH(true, true, (sgemm_dw_grouped_kernel<32, 64, false>), dim3((unsigned)first), dim3(256), 0, s, tab);
  return s2f_check_launch("s2f_spike_gemm_dw_grouped");
}

extern "C" int s2f_spike_gemm_fwd_bf16_ex(const uint16_t* a_split, int64_t a_batch_stride, const uint16_t* X,
                                          int64_t x_batch_stride, int k_inner, int64_t x_outer_stride, const float* bias,
                                          int64_t bias_batch_stride, float out_scale, float* Y, int batch, int M, int N, int K,
                                          int Mpad, int Kpad, void* stream) {
  S2F_REQUIRE(k_inner > 0 && k_inner % 32 == 0 && K % k_inner == 0, S2F_EINVAL,
              "s2f_spike_gemm_fwd_bf16_ex: k_inner must be a multiple of 32 that divides K");
  S2F_REQUIRE((a_batch_stride & 7) == 0 && (x_batch_stride & 7) == 0 && (x_outer_stride & 7) == 0, S2F_EALIGN,
              "s2f_spike_gemm_fwd_bf16_ex: strides must keep 16-byte alignment");
  return fwd_launch("s2f_spike_gemm_fwd_bf16_ex", a_split, X, bias, Y, batch, M, N, K, Mpad, Kpad, 3, false, Conv3{0, 0, 0}, stream,
                    GemmEx{1, k_inner, a_batch_stride, x_batch_stride, x_outer_stride, bias_batch_stride, out_scale});
}
