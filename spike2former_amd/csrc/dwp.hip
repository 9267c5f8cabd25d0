// Weight gradient of a spike-fed convolution on the LDS-DMA pipeline, 8 wavefronts in two halves (gfx950, round 5).
//
//   dW[m][k] += sum_b sum_l dY[b][m][l] X[b][k][l]        dY fp32 [B][M][L],  X bf16 spikes [B][K][L],  L % 32 == 0
//
// The round-2 kernel (gemm_bf16.hip: dw_tile_body) stages BOTH operands through registers into a 64 x 128 tile and its four
// wavefronts run  stage -> barrier -> fragment reads -> MFMA -> barrier  one after the other: per step a workgroup moves
// 40 KB into LDS with ds_write_b64 and 112 KB out of it for 24 MFMAs per wavefront -- more LDS cycles than MFMA cycles
// (MFMA busy 0.27 in the step, profiles/r04_pmc_mfma.txt).  Here:
//   * tile 128 (dY rows) x 256 (X rows), contraction step 32, eight wavefronts of 64 x 64: one fragment read feeds 2 MFMAs per
//     term (8 ds_read_b128 for 12 MFMAs), LDS traffic per MFMA a third of the old tile's;
//   * X -- contraction-contiguous bf16 -- lands in LDS by global_load_lds_dwordx4 (no registers, no ds_write): one copy
//     instruction = 16 rows x 64 bytes, the 16-byte chunk c of row r stored at c ^ ((r >> 2) & 3) (the swizzle is applied to the
//     per-lane SOURCE address: an LDS-DMA lands lane-linear), four ring slots, issued two tiles ahead;
//   * dY is split hi + mid + lo while it passes through registers (it is fp32 and needs the VALU) into two stages of three
//     64-byte-row planes with the same swizzle;
//   * the two halves of the workgroup (wavefronts 0-3 / 4-7: one wavefront of each SIMD) run in OPPOSITE phase, one barrier per
//     phase (MI355X_MICROARCH.md "Two waves per SIMD"): while one half multiplies tile t (16 fragment reads, 24 MFMAs per
//     wavefront) the other stages its share of tile t + 1 (split + 6 ds_write_b64), loads its share of dY(t + 2) and issues its
//     copies of X(t + 2); then they swap.  On every SIMD one wavefront is in its MFMA segment while its partner is in its
//     memory segment; the younger half runs at s_setprio 1 throughout (no per-segment flips).
//         half 0:  C(0) | S(1) | C(1) | S(2) | ...          half 1:  S(1) | C(0) | S(2) | C(1) | ...
// Rows past M / K are read from clamped addresses and never stored (an output row depends on its own operand row only).
// Split-K over B * L with fp32 atomics into dW, as the kernel it replaces.  Reference call sites: the autograd weight gradients
// of every 1x1 Conv2d / Conv1d fed by a Q_IFNode (mmseg/models/backbones/sdtv2.py:222-255, 304-306;
// mmcv_spike/transformer.py:213-236, 758-763; mmdet/models/layers/pixel_decoder.py:368-404).
#include "gemm_common.h"
#include <cstdlib>

#pragma clang fp contract(fast)

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// LDS-DMA as a BUFFER load (buffer_load_dwordx4 ... lds), not global_load_lds: the latter is FLAT-encoded and touches two address
// spaces, which makes hipcc's wait-count pass treat every later vector-memory dependency as "pending FLAT" and wait vmcnt(0)
// for it -- the dY registers of tile u could then not be consumed while the group of tile u + 1 stays in flight.  With the buffer
// form it counts (`s_waitcnt vmcnt(4)` in front of the split).  base: wave-uniform, off: per-lane byte offset (< 2^32).
__device__ __forceinline__ void dma16(const void* base, unsigned off, void* lds_dst) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, off, 0, 0, 0);
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_b128_asm(unsigned byte_addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(byte_addr), "n"(OFF));
  return r;
}
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void mfma_bf16(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long)(lds_void*)p; }
__device__ __forceinline__ void phase_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int TM = 128, TK = 256, BL = 32;
constexpr int A_PLANE = TM * 64;                 // one term plane: 128 rows x 64 bytes
constexpr int A_STAGE = 3 * A_PLANE;             // 24 KiB
constexpr int B_STAGE = TK * 64;                 // 16 KiB
constexpr int NA = 2, NB = 4;
constexpr int LDS_BYTES = NA * A_STAGE + NB * B_STAGE;          // 112 KiB: one workgroup per CU

// SYM = true: the symmetric schedule for comparison (all eight wavefronts stage, barrier, multiply, barrier)
// KO (probe builds only, S2F_DWP_PROBE): 1 = no MFMAs, 2 = no fragment reads, 4 = no loads / copies after the prologue, 8 = no staging
template <bool SYM, int KO = 0>
__device__ __forceinline__ void dwp_body(const float* __restrict__ dY, const unsigned short* __restrict__ X,
                                         float* __restrict__ dW, int M, int K, int L, int total_steps, int sps, int k_tiles,
                                         int tile, int split) {
  // two OBJECTS: hipcc orders an LDS store behind every LDS-DMA in flight that it cannot prove disjoint (s_waitcnt vmcnt(0) in front
  // of the ds_write of the dY planes, i.e. the X copies of the next tile drained every step when both lived in one array)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NA * A_STAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char smem_ring[NB * B_STAGE];
  const int s_begin = split * sps;
  const int n = min(total_steps, s_begin + sps) - s_begin;
  if (n <= 0) return;
  const int m0 = (tile / k_tiles) * TM, k0 = (tile % k_tiles) * TK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, wn = wave & 3, wm = half;

  // ---- staging shares: each half stages 64 rows of dY (2 float4 per thread) and 8 of the 16 X copies (2 per wavefront)
  const int ht = tid & 255;
  unsigned oa[2], aw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = ht + i * 256;
    const int row = half * 64 + (c >> 3), q4 = c & 7;
    oa[i] = ((unsigned)min(m0 + row, M - 1) * (unsigned)L + (unsigned)(q4 * 4)) * 4u;          // BYTES (M L < 2^30)
    aw[i] = (unsigned)(row * 64 + ((((q4 >> 1) ^ ((row >> 2) & 3))) << 4) + (q4 & 1) * 8);
  }
  unsigned ob[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = (wave * 2 + q) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    ob[q] = ((unsigned)min(k0 + r, K - 1) * (unsigned)L + (unsigned)(c * 8)) * 2u;          // BYTES (K L < 2^31)
  }
  // ---- fragment addresses (bytes inside a stage)
  unsigned aoff[2][2], boff[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + (lane & 31), rb = wn * 64 + i * 32 + (lane & 31);
      const int c = ks * 2 + (lane >> 5);
      aoff[ks][i] = (unsigned)(ra * 64 + ((c ^ ((ra >> 2) & 3)) << 4));
      boff[ks][i] = (unsigned)(rb * 64 + ((c ^ ((rb >> 2) & 3)) << 4));
    }
  const unsigned smem_a = lds_addr(smem), smem_b = lds_addr(smem_ring);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // cursor of the next tile to LOAD (wave-uniform): batch element, contraction offset
  const int lsteps = L >> 5;
  int cb = s_begin / lsteps, cl = (s_begin - cb * lsteps) << 5;
  int next_slot = 0, loaded = 0;
  f32x4 areg0[2], areg1[2];                    // dY shares of the even / odd tiles in flight (two tiles ahead)
  auto load_next = [&](f32x4 (&areg)[2]) __attribute__((always_inline)) {
    // wave-uniform base + loop-invariant 32-bit byte offset per lane
    const char* pa = reinterpret_cast<const char*>(dY + (int64_t)cb * M * L + cl);
    const char* px = reinterpret_cast<const char*>(X + (int64_t)cb * K * L + cl);
#pragma unroll
    for (int i = 0; i < 2; ++i) areg[i] = *reinterpret_cast<const f32x4*>(pa + oa[i]);
#pragma unroll
    for (int q = 0; q < 2; ++q) dma16(px, ob[q], smem_ring + next_slot * B_STAGE + (wave * 2 + q) * 1024);
    next_slot = next_slot == NB - 1 ? 0 : next_slot + 1;
    // Past the last tile the cursor stays where it is: the requests are issued UNCONDITIONALLY (a re-read of the last tile into
    // registers nobody consumes and a ring slot nobody reads again) -- a conditional request makes hipcc count the outstanding
    // operations of the path WITHOUT it, i.e. wait for part of the newest group in the steady state.
    if (++loaded < n) {
      cl += BL;
      if (cl == L) {
        cl = 0;
        ++cb;
      }
    }
  };
  // S(u): this thread's share of dY tile u (held in `areg`) goes into stage u & 1; then tile u + 2 is requested into the same
  // registers and ring slot (u + 2) % 4.  In flight at the head of S(u): the group {2 loads, 2 copies} of tile u + 1 (allowed to
  // stay in flight: vmcnt(4)) behind the group of tile u, which must have landed -- it was issued two steps ago.
  auto stage = [&](int u, f32x4 (&areg)[2]) __attribute__((always_inline)) {
    if (u >= n || (KO & 8)) return;
    if (u + 1 < n && !(KO & 4)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned char* as = smem + (u & 1) * A_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4 v = areg[i];
      unsigned h0, m0_, l0_, h1, m1, l1;
      s2f_split3x2(v.x, v.y, h0, m0_, l0_);
      s2f_split3x2(v.z, v.w, h1, m1, l1);
      *reinterpret_cast<u32x2*>(as + aw[i]) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(as + A_PLANE + aw[i]) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(as + 2 * A_PLANE + aw[i]) = u32x2{l0_, l1};
    }
    if (!(KO & 4)) load_next(areg);
  };
  // C(t): 16 fragment reads requested at once, the MFMAs of k slice 0 run under the landing of slice 1's fragments
  auto compute = [&](int t, int slot) __attribute__((always_inline)) {
    const unsigned ab = smem_a + (t & 1) * A_STAGE, bb = smem_b + slot * B_STAGE;
    bf16x8 bf[2][2], af[2][3][2];
    if (KO & 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          asm volatile("" : "=v"(bf[ks][j]));
#pragma unroll
          for (int t3 = 0; t3 < 3; ++t3) asm volatile("" : "=v"(af[ks][t3][j]));
        }
    } else
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[ks][j] = lds_b128_asm<0>(bb + boff[ks][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[ks][0][i] = lds_b128_asm<0>(ab + aoff[ks][i]);
        af[ks][1][i] = lds_b128_asm<A_PLANE>(ab + aoff[ks][i]);
        af[ks][2][i] = lds_b128_asm<2 * A_PLANE>(ab + aoff[ks][i]);
      }
    }
    lds_wait<8>();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1) lds_wait<0>();
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (KO & 1) asm volatile("" ::"v"(af[ks][t3][i]), "v"(bf[ks][j]));
            else mfma_bf16(acc[i][j], af[ks][t3][i], bf[ks][j]);
          }
    }
  };

  // prologue: tiles 0 and 1 requested, tile 0 staged by everybody (which requests tile 2)
  load_next(areg0);
  load_next(areg1);
  stage(0, areg0);
  if (half == 1 && !SYM) asm volatile("s_setprio 1");
  phase_barrier();
  int slot = 0;
#define S2F_NEXT_SLOT slot = slot == NB - 1 ? 0 : slot + 1
  // Both loops START with the staging segment: hipcc waits for the dY registers at the loop header, which must be the point
  // where they are consumed -- two steps after their loads were issued -- not the head of a multiply segment.  Unrolled by two:
  // the register set of a tile is chosen by its parity.
  if (SYM || half == 0) {
    compute(0, 0);
    slot = 1;
    phase_barrier();
    for (int k = 1; k < n; k += 2) {
      stage(k, areg1);
      phase_barrier();
      compute(k, slot);
      S2F_NEXT_SLOT;
      phase_barrier();
      if (k + 1 < n) {
        stage(k + 1, areg0);
        phase_barrier();
        compute(k + 1, slot);
        S2F_NEXT_SLOT;
        phase_barrier();
      }
    }
  } else {
    for (int k = 0; k < n; k += 2) {
      stage(k + 1, areg1);
      phase_barrier();
      compute(k, slot);
      S2F_NEXT_SLOT;
      if (k < n - 1) {
        phase_barrier();
        stage(k + 2, areg0);
        phase_barrier();
        compute(k + 1, slot);
        S2F_NEXT_SLOT;
        if (k + 1 < n - 1) phase_barrier();
      }
    }
  }
#undef S2F_NEXT_SLOT
  if (half == 1 && !SYM) asm volatile("s_setprio 0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may land after this workgroup has released its LDS
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[i][j]));
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = k0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < K) atomicAdd(dW + (int64_t)row * K + col, acc[i][j][r]);
      }
    }
}

// workgroup f runs on XCD f % 8: give each XCD a contiguous range of (split, tile) ids, tile fastest -- the tiles of one
// contraction split stream the same dY / X slices at the same time and share them through that XCD's L2 (see gemm_bf16.hip)
__device__ __forceinline__ int xcd_contiguous(int f, int total) {
  const int chunk = total >> 3, rem = total & 7;
  const int xcd = f & 7, idx = f >> 3;
  return xcd * chunk + min(xcd, rem) + idx;
}

template <bool SYM, int KO = 0>
__global__ __launch_bounds__(512, 1) void dwp_kernel(const float* __restrict__ dY, const unsigned short* __restrict__ X,
                                                     float* __restrict__ dW, int M, int K, int L, int total_steps, int sps,
                                                     int k_tiles, int tiles) {
  const int id = xcd_contiguous(blockIdx.x, gridDim.x);
  dwp_body<SYM, KO>(dY, X, dW, M, K, L, total_steps, sps, k_tiles, id % tiles, id / tiles);
}

constexpr int kMaxJobs = 56;
struct DwpJob {
  const float* dY;
  const unsigned short* X;
  float* dW;
  int M, K, L, total_steps;
  int first_block, sps, k_tiles, tiles;
};
struct DwpJobTable {
  int njobs;
  DwpJob job[kMaxJobs];
};

template <bool SYM, int KO = 0>
__global__ __launch_bounds__(512, 1) void dwp_grouped_kernel(const DwpJobTable tab) {
  const int id = xcd_contiguous(blockIdx.x, gridDim.x);
  int lo = 0, hi = tab.njobs - 1;                         // last job whose first block <= id (wave-uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab.job[mid].first_block <= id) lo = mid; else hi = mid - 1;
  }
  const DwpJob& j = tab.job[lo];
  const int local = id - j.first_block;
  dwp_body<SYM, KO>(j.dY, j.X, j.dW, j.M, j.K, j.L, j.total_steps, j.sps, j.k_tiles, local % j.tiles, local / j.tiles);
}

bool dwp_shape_ok(int batch, int M, int K, int L) {
  return batch > 0 && M > 0 && K > 0 && L >= 32 && (L & 31) == 0 && (int64_t)M * L < (1ll << 30) && (int64_t)K * L < (1ll << 31) &&
         (int64_t)batch * (L >> 5) < (1ll << 30);
}

}  // namespace

// 1 when the pipelined kernel takes this shape (the host's dispatch asks before choosing it)
extern "C" int s2f_spike_gemm_dw_pipe_ok(int batch, int M, int K, int L) { return dwp_shape_ok(batch, M, K, L) ? 1 : 0; }

// dW (+)= sum_b dY[b] X[b]^T on the pipelined kernel.  cfg: 0 = two-halves schedule, 1 = symmetric schedule (probe);
// target_wgs <= 0: the default number of workgroups the contraction is split for.
extern "C" int s2f_spike_gemm_dw_pipe(const float* dY, const uint16_t* X, float* dW, int batch, int M, int K, int L,
                                      int accumulate, int cfg, int target_wgs, void* stream) {
  S2F_REQUIRE(dY && X && dW, S2F_EINVAL, "s2f_spike_gemm_dw_pipe: null pointer");
  S2F_REQUIRE(dwp_shape_ok(batch, M, K, L), S2F_EINVAL, "s2f_spike_gemm_dw_pipe: needs L %% 32 == 0, M L < 2^30, K L < 2^31 (M=%d K=%d L=%d)",
              M, K, L);
  S2F_REQUIRE(s2f_aligned16(dY) && s2f_aligned16(X), S2F_EALIGN, "s2f_spike_gemm_dw_pipe: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && s2f_zero_async(dW, sizeof(float) * (size_t)M * K, s) != S2F_OK)
    return s2f_check_launch("s2f_spike_gemm_dw_pipe memset");
  const int k_tiles = (K + TK - 1) / TK, tiles = ((M + TM - 1) / TM) * k_tiles;
  const int total = batch * (L >> 5);
  if (target_wgs <= 0) target_wgs = 768;
  int sps = (int)(((int64_t)total * tiles + target_wgs - 1) / target_wgs);
  if (sps < 8) sps = 8;
  if (sps > total) sps = total;
  const int splits = (total + sps - 1) / sps;
  const dim3 grid((unsigned)(tiles * splits));
  if (cfg == 1)
    S2F_LAUNCH(true, true, (dwp_kernel<true>), grid, dim3(512), 0, s, dY, X, dW, M, K, L, total, sps, k_tiles, tiles);
  else
    S2F_LAUNCH(true, true, (dwp_kernel<false>), grid, dim3(512), 0, s, dY, X, dW, M, K, L, total, sps, k_tiles, tiles);
  return s2f_check_launch("s2f_spike_gemm_dw_pipe");
}

// MANY weight gradients in one launch (what s2f_spike_gemm_dw_grouped is to the round-2 kernel).  jobs (HOST array):
// njobs x {dY, X, dW (pointers), batch, M, K, L}; every dW is accumulated into.
extern "C" int s2f_spike_gemm_dw_pipe_grouped(const int64_t* jobs, int njobs, int cfg, int target_wgs, void* stream) {
  S2F_REQUIRE(jobs && njobs > 0 && njobs <= kMaxJobs, S2F_EINVAL, "s2f_spike_gemm_dw_pipe_grouped: 1 .. %d jobs", kMaxJobs);
  DwpJobTable tab;
  tab.njobs = njobs;
  int64_t work = 0;
  for (int i = 0; i < njobs; ++i) {
    const int64_t* r = jobs + 7 * i;
    DwpJob& j = tab.job[i];
    j.dY = reinterpret_cast<const float*>(r[0]);
    j.X = reinterpret_cast<const unsigned short*>(r[1]);
    j.dW = reinterpret_cast<float*>(r[2]);
    const int B = (int)r[3];
    j.M = (int)r[4], j.K = (int)r[5], j.L = (int)r[6];
    S2F_REQUIRE(j.dY && j.X && j.dW && dwp_shape_ok(B, j.M, j.K, j.L), S2F_EINVAL, "s2f_spike_gemm_dw_pipe_grouped: bad job %d", i);
    S2F_REQUIRE(s2f_aligned16(j.dY) && s2f_aligned16(j.X), S2F_EALIGN, "s2f_spike_gemm_dw_pipe_grouped: job %d misaligned", i);
    j.k_tiles = (j.K + TK - 1) / TK;
    j.tiles = ((j.M + TM - 1) / TM) * j.k_tiles;
    j.total_steps = B * (j.L >> 5);
    work += (int64_t)j.tiles * j.total_steps;
  }
  if (target_wgs <= 0) target_wgs = 768;
  int sps = (int)((work + target_wgs - 1) / target_wgs);
  if (sps < 8) sps = 8;
  int64_t first = 0;
  for (int i = 0; i < njobs; ++i) {
    DwpJob& j = tab.job[i];
    j.sps = sps < j.total_steps ? sps : j.total_steps;
    const int splits = (j.total_steps + j.sps - 1) / j.sps;
    j.first_block = (int)first;
    first += (int64_t)j.tiles * splits;
  }
  S2F_REQUIRE(first < (1ll << 31), S2F_EINVAL, "s2f_spike_gemm_dw_pipe_grouped: grid too large");
  hipStream_t s = (hipStream_t)stream;
#ifdef S2F_DWP_PROBE
#define S2F_KO(N) if (cfg == 2 * N + 1) { S2F_LAUNCH(true, true, (dwp_grouped_kernel<true, N>), dim3((unsigned)first), dim3(512), 0, s, tab); return s2f_check_launch("ko"); } \
                  if (cfg == 2 * N) { S2F_LAUNCH(true, true, (dwp_grouped_kernel<false, N>), dim3((unsigned)first), dim3(512), 0, s, tab); return s2f_check_launch("ko"); }
  S2F_KO(1) S2F_KO(2) S2F_KO(3) S2F_KO(4) S2F_KO(8) S2F_KO(12) S2F_KO(13) S2F_KO(15)
#undef S2F_KO
#endif
  if (cfg == 1)
    S2F_LAUNCH(true, true, (dwp_grouped_kernel<true>), dim3((unsigned)first), dim3(512), 0, s, tab);
  else
    S2F_LAUNCH(true, true, (dwp_grouped_kernel<false>), dim3((unsigned)first), dim3(512), 0, s, tab);
  return s2f_check_launch("s2f_spike_gemm_dw_pipe_grouped");
}
