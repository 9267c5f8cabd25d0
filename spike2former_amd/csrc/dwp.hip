// Weight gradient of a spike-fed convolution on the LDS-DMA pipeline, 8 wavefronts in two halves (gfx950, round 5).
//
//   dW[m][k] += sum_b sum_l dY[b][m][l] X[b][k][l]        dY fp32 [B][M][L],  X bf16 spikes [B][K][L],  L % 4 == 0
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
// the same with the hardware's range check: `off` is the byte offset from the START of a tensor of `bytes` bytes; a lane whose
// offset falls outside (a tap above the first / below the last image row of the first / last plane: the unsigned offset wraps)
// lands zeros instead of touching memory
__device__ __forceinline__ void dma16_checked(const void* tensor, unsigned bytes, unsigned off, void* lds_dst) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(tensor), 0, bytes, 0x00020000);
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
constexpr int RAW_SLOT = TM * BL * 4;            // fp32 dY tile as it arrives: 16 KiB (2 KiB per wavefront)
constexpr int NRAW = 3;

// SYM = false: two halves in opposite phase (planes double-buffered, four ring slots: 160 KiB of LDS);
// SYM = true : every wavefront in the same phase  stage | barrier | multiply | barrier  (one plane stage, three ring slots: 120 KiB)
// KO (probe builds only, S2F_DWP_PROBE): 1 = no MFMAs, 2 = no fragment reads, 4 = no copies after the prologue, 8 = no staging
// CONV: the implicit 3x3 convolution (stride 1, padding 1).  Row k = tap * C + c of the virtual im2col matrix is channel plane c
// shifted by the tap; X is the activation [B][C][H][W] itself and Xs a copy shifted by ONE element (Xs[i] = X[i + 1]), so that
// the horizontal taps are dword-aligned copy sources too: kx = 1 reads X at l + (ky - 1) W, kx = 2 reads Xs at the same place,
// kx = 0 reads Xs two elements earlier.  What the zero padding would have supplied is zeroed in the X fragments: a tap above /
// below the image (whole 8-pixel chunk: W % 8 == 0), the first pixel of a row for kx = 0, the last for kx = 2.  A copy instruction covers 16 rows of one tap (C % 16 == 0).
struct DwpConv {
  const unsigned short* Xs;
  int C, H, W;
  unsigned x_bytes;          // bytes of X (= of Xs): B C H W 2 < 2^31
  int wl;                    // 1: dW in the weight's layout [M][C][3][3] (column (tap, c) of the tap-major product -> c * 9 + tap)
};
// RAG ("ragged"): L % 32 != 0 (L % 4 == 0) -- the 100-token layers of the decoder, the 50 x 84 maps of C5.  The last step of a batch
// element runs past the end of the rows: the copies are range-checked against the whole tensors (B M L < 2^30, B K L < 2^31: the
// offsets count from the start of the tensor), dY is zeroed past l = L while it is split -- whatever X holds there (the head of
// the next row: finite spikes, or the zeros of the range check) is multiplied by zero.
template <bool SYM, int KO = 0, bool CONV = false, bool RAG = false>
__device__ __forceinline__ void dwp_body(const float* __restrict__ dY, const unsigned short* __restrict__ X,
                                         float* __restrict__ dW, int M, int K, int L, int k_tiles, int tile, int s_begin,
                                         int n, DwpConv cv = DwpConv{}, int B = 0) {
  static_assert(!(CONV && RAG), "the ragged form is for the plain products");
  constexpr int NA = SYM ? 1 : 2, NB = SYM ? 3 : 4;
  // separate OBJECTS: hipcc orders an LDS store behind every LDS-DMA in flight that it cannot prove disjoint
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NA * A_STAGE];          // dY planes hi | mid | lo
  __shared__ __attribute__((aligned(1024))) unsigned char smem_ring[NB * B_STAGE];     // X tiles
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[NRAW * RAW_SLOT];   // dY tiles as fp32, one 16-byte mailbox per lane
  if (n <= 0) return;
  const int m0 = (tile / k_tiles) * TM, k0 = (tile % k_tiles) * TK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, wn = wave & 3, wm = half;

  // ---- this wavefront's share of a tile: 16 rows of dY (two copies of 8 rows x 128 bytes: lane -> row lane >> 3, float4
  // lane & 7, landing in mailbox `lane` of the copy) and 32 rows of X (two copies of 16 rows x 64 bytes)
  unsigned oa[2], aw[2], ob[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 16 + i * 8 + (lane >> 3), q4 = lane & 7;
    oa[i] = ((unsigned)min(m0 + row, M - 1) * (unsigned)L + (unsigned)(q4 * 4)) * 4u;          // BYTES (M L < 2^30)
    aw[i] = (unsigned)(row * 64 + ((((q4 >> 1) ^ ((row >> 2) & 3))) << 4) + (q4 & 1) * 8);
    const int r = (wave * 2 + i) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    if constexpr (CONV) {
      const int k = min(k0 + r, K - 1), tap = k / cv.C, ch = k - tap * cv.C, ky = tap / 3, kx = tap - 3 * ky;
      // byte offset from the start of the tensor, without the batch element and the step (added per request); may be negative
      ob[i] = (unsigned)((ch * L + (ky - 1) * cv.W + (kx == 0 ? -2 : 0) + c * 8) * 2);
    } else {
      ob[i] = ((unsigned)min(k0 + r, K - 1) * (unsigned)L + (unsigned)(c * 8)) * 2u;             // BYTES (K L < 2^31)
    }
  }
  // CONV: which tensor each of this wavefront's two copy instructions reads (wave-uniform), and per X-fragment row block the tap
  bool shifted[2] = {false, false};
  int fky[2] = {1, 1}, fkx[2] = {1, 1};
  unsigned x_bytes = 0;
  if constexpr (CONV) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tap = min(k0 + (wave * 2 + i) * 16, K - 1) / cv.C;
      shifted[i] = (tap % 3) != 1;
      const int kf = min(k0 + wn * 64 + i * 32 + (lane & 31), K - 1), tf = kf / cv.C;
      fky[i] = tf / 3, fkx[i] = tf - 3 * fky[i];
    }
    x_bytes = cv.x_bytes;
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
  const unsigned raw_mine = lds_addr(smem_raw) + wave * 2048 + lane * 16;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // cursor of the next tile to REQUEST (wave-uniform): batch element, contraction offset, ring positions
  const int lsteps = RAG ? (L + 31) >> 5 : L >> 5;
  int cb = s_begin / lsteps, cl = (s_begin - cb * lsteps) << 5;
  const unsigned dy_bytes = RAG ? (unsigned)B * (unsigned)M * (unsigned)L * 4u : 0u;
  const unsigned xr_bytes = RAG ? (unsigned)B * (unsigned)K * (unsigned)L * 2u : 0u;
  int next_slot = 0, next_raw = 0, loaded = 0;
  // one group = 4 copies per wavefront: {dY rows 0-7, dY rows 8-15, X rows 0-15, X rows 16-31} of its share
  auto request = [&]() __attribute__((always_inline)) {
    if constexpr (RAG) {
      const unsigned ua = (unsigned)(cb * M * L + cl) * 4u;
#pragma unroll
      for (int i = 0; i < 2; ++i) dma16_checked(dY, dy_bytes, oa[i] + ua, smem_raw + next_raw * RAW_SLOT + wave * 2048 + i * 1024);
    } else {
      const char* pa = reinterpret_cast<const char*>(dY + (int64_t)cb * M * L + cl);
#pragma unroll
      for (int i = 0; i < 2; ++i) dma16(pa, oa[i], smem_raw + next_raw * RAW_SLOT + wave * 2048 + i * 1024);
    }
    if constexpr (RAG) {
      const unsigned ux = (unsigned)(cb * K * L + cl) * 2u;
#pragma unroll
      for (int q = 0; q < 2; ++q) dma16_checked(X, xr_bytes, ob[q] + ux, smem_ring + next_slot * B_STAGE + (wave * 2 + q) * 1024);
    } else if constexpr (CONV) {
      const unsigned uni = (unsigned)((cb * cv.C * L + cl) * 2);          // batch element + step, bytes (x_bytes < 2^31)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        dma16_checked(shifted[q] ? cv.Xs : X, shifted[q] ? x_bytes + 32u : x_bytes, ob[q] + uni + (shifted[q] ? 16u : 0u),
                      smem_ring + next_slot * B_STAGE + (wave * 2 + q) * 1024);          // (Xs: 8 elements of front pad)
    } else {
      const char* px = reinterpret_cast<const char*>(X + (int64_t)cb * K * L + cl);
#pragma unroll
      for (int q = 0; q < 2; ++q) dma16(px, ob[q], smem_ring + next_slot * B_STAGE + (wave * 2 + q) * 1024);
    }
    next_slot = next_slot == NB - 1 ? 0 : next_slot + 1;
    next_raw = next_raw == NRAW - 1 ? 0 : next_raw + 1;
    // past the last tile the cursor stays: the requests are issued unconditionally (into slots nobody reads again), so that the
    // number of copies in flight at any point of the loop does not depend on the path taken
    if (++loaded < n) {
      cl += BL;
      if (cl >= L) {
        cl = 0;
        ++cb;
      }
    }
  };
  // S(u): this wavefront's 16 rows of dY tile u -- landed in its own mailboxes, so no barrier is needed between the copy and the
  // conversion -- are split hi + mid + lo into plane stage u % NA; then tile u + 2 is requested.  In flight at the head of S(u):
  // group u + 1 (4 copies: may stay in flight) behind group u.
  int raw_slot = 0;
  int sl = cl + (lane & 7) * 4;          // RAG: contraction offset of this lane's float4 in the tile being staged
  auto stage = [&](int u) __attribute__((always_inline)) {
    if (u >= n || (KO & 8)) return;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    const unsigned rb_ = raw_mine + raw_slot * RAW_SLOT;
    raw_slot = raw_slot == NRAW - 1 ? 0 : raw_slot + 1;
    f32x4 v[2];
    asm volatile("ds_read_b128 %0, %1" : "=v"(v[0]) : "v"(rb_));
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(v[1]) : "v"(rb_));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1])::"memory");
    if constexpr (RAG) {
      if (sl >= L) v[0] = v[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      sl += BL;
      if (sl - (int)(lane & 7) * 4 >= L) sl = (lane & 7) * 4;
    }
    unsigned char* as = smem + (NA == 1 ? 0 : (u & 1)) * A_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned h0, m0_, l0_, h1, m1, l1;
      if (KO & 16) {
        h0 = __float_as_uint(v[i].x), h1 = __float_as_uint(v[i].y), m0_ = __float_as_uint(v[i].z), m1 = __float_as_uint(v[i].w), l0_ = h0, l1 = m1;
      } else {
        s2f_split3x2(v[i].x, v[i].y, h0, m0_, l0_);
        s2f_split3x2(v[i].z, v[i].w, h1, m1, l1);
      }
      if (KO & 32) {
        asm volatile("" ::"v"(h0), "v"(h1), "v"(m0_), "v"(m1), "v"(l0_), "v"(l1));
      } else {
        *reinterpret_cast<u32x2*>(as + aw[i]) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(as + A_PLANE + aw[i]) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(as + 2 * A_PLANE + aw[i]) = u32x2{l0_, l1};
      }
    }
    if (!(KO & 4)) request();
  };
  // CONV: image row / first column of the tile being MULTIPLIED (wave-uniform)
  int cy = 0, cx = 0;
  if constexpr (CONV) {
    const int l_first = (s_begin % lsteps) << 5;
    cy = l_first / cv.W, cx = l_first - cy * cv.W;
  }
  // C(t): 16 fragment reads requested at once, the MFMAs of k slice 0 run under the landing of slice 1's fragments
  auto compute = [&](int t, int slot) __attribute__((always_inline)) {
    const unsigned ab = smem_a + (NA == 1 ? 0 : (t & 1)) * A_STAGE, bb = smem_b + slot * B_STAGE;
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
    if constexpr (CONV) {
      // the waits take the X fragments as in / out operands: the zeroing below must not be scheduled above them
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bf[0][0]), "+v"(bf[0][1])::"memory");
    } else {
      lds_wait<8>();
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1) {
        if constexpr (CONV) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[1][0]), "+v"(bf[1][1])::"memory");
        else lds_wait<0>();
      }
      if constexpr (CONV) {
        // first column / image row of this lane's 8-pixel chunk: a 32-pixel step may run into the next image row (W % 8 == 0 keeps a
        // chunk inside one row, H W % 32 == 0 a step inside one image)
        int xc = cx + (ks * 2 + (lane >> 5)) * 8, yc = cy;
        if (xc >= cv.W) xc -= cv.W, ++yc;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int yy = yc + fky[j] - 1;
          u32x4 v = *reinterpret_cast<u32x4*>(&bf[ks][j]);
          const bool vok = yy >= 0 && yy < cv.H;
          v.x = (vok && !(fkx[j] == 0 && xc == 0)) ? v.x : (vok ? (v.x & 0xffff0000u) : 0u);
          v.y = vok ? v.y : 0u;
          v.z = vok ? v.z : 0u;
          v.w = (vok && !(fkx[j] == 2 && xc + 8 == cv.W)) ? v.w : (vok ? (v.w & 0x0000ffffu) : 0u);
          *reinterpret_cast<u32x4*>(&bf[ks][j]) = v;
        }
      }
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
    if constexpr (CONV) {
      cx += BL;
      if (cx >= cv.W) {
        cx -= cv.W;
        cy = cy + 1 == cv.H ? 0 : cy + 1;
      }
    }
  };

  // prologue: tiles 0 and 1 requested, tile 0 staged by everybody (which requests tile 2)
  request();
  request();
  stage(0);
  if (half == 1 && !SYM) asm volatile("s_setprio 1");
  phase_barrier();
  int slot = 0;
#define S2F_NEXT_SLOT slot = slot == NB - 1 ? 0 : slot + 1
  if (SYM) {
    // the planes have ONE stage: S(k + 1) follows the barrier behind C(k)
    for (int k = 0; k < n; ++k) {
      compute(k, slot);
      S2F_NEXT_SLOT;
      phase_barrier();
      stage(k + 1);
      phase_barrier();
    }
  } else if (half == 0) {
    // half 0:  C(0) | S(1) | C(1) | S(2) | ...          half 1:  S(1) | C(0) | S(2) | C(1) | ...
    compute(0, 0);
    slot = 1;
    phase_barrier();
    for (int k = 1; k < n; ++k) {
      stage(k);
      phase_barrier();
      compute(k, slot);
      S2F_NEXT_SLOT;
      phase_barrier();
    }
  } else {
    for (int k = 0; k < n; ++k) {
      stage(k + 1);
      phase_barrier();
      compute(k, slot);
      S2F_NEXT_SLOT;
      if (k < n - 1) phase_barrier();
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
      int colw = col;
      if (CONV && cv.wl) {
        const int tap = col / cv.C;
        colw = (col - tap * cv.C) * 9 + tap;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < K) atomicAdd(dW + (int64_t)row * K + colw, acc[i][j][r]);
      }
    }
}

// workgroup f runs on XCD f % 8: give each XCD a contiguous range of the linearised work, so that the workgroups of one XCD stream
// neighbouring tiles of the same job (shared dY rows / X rows) through that XCD's L2 (see gemm_bf16.hip)
__device__ __forceinline__ int xcd_contiguous(int f, int total) {
  const int chunk = total >> 3, rem = total & 7;
  const int xcd = f & 7, idx = f >> 3;
  return xcd * chunk + min(xcd, rem) + idx;
}

// Work decomposition ("stream-K"): the (tile, contraction step) pairs of all jobs are laid out in one line -- job by job, tile by
// tile, a tile's steps consecutive -- and workgroup w takes the w-th of gridDim.x EQUAL pieces of it.  A piece is at most a tail
// of one tile, some whole tiles, a head of another: the accumulators are flushed (fp32 atomics) when the tile changes, i.e. one
// flush per (workgroup, tile touched) instead of one per fixed-length contraction split -- the 128 KiB atomic tile of a
// 768-workgroup split cost 88 us of a 400 us launch (tools/probe_dwp_ko.py: "barriers + epilogue only"); every CU gets the same
// amount of work, there is no second, partly filled round.
constexpr int kMaxJobs = 56;
struct DwpJob {
  const float* dY;
  const unsigned short* X;
  float* dW;
  int M, K, L, steps;          // steps = batch * ceil(L / 32) contraction steps per tile
  int first_work, k_tiles;     // first_work: index of this job's first (tile, step) pair in the line
};
struct DwpJobTable {
  int njobs, work, quota;      // work: total (tile, step) pairs; quota: pairs per workgroup
  DwpJob job[kMaxJobs];
};

template <bool SYM, int KO = 0, bool RAG = false>
__global__ __launch_bounds__(512, 1) void dwp_grouped_kernel(const DwpJobTable tab) {
  const int id = xcd_contiguous(blockIdx.x, gridDim.x);
  int w0 = id * tab.quota;
  const int w1 = min(tab.work, w0 + tab.quota);
  while (w0 < w1) {
    int lo = 0, hi = tab.njobs - 1;                       // last job whose first pair <= w0 (wave-uniform)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tab.job[mid].first_work <= w0) lo = mid; else hi = mid - 1;
    }
    const DwpJob& j = tab.job[lo];
    const int local = w0 - j.first_work;
    const int tile = local / j.steps, s = local - tile * j.steps;
    const int n = min(j.steps - s, w1 - w0);
    dwp_body<SYM, KO, false, RAG>(j.dY, j.X, j.dW, j.M, j.K, j.L, j.k_tiles, tile, s, n, DwpConv{}, RAG ? j.steps / ((j.L + 31) >> 5) : 0);
    w0 += n;
    __syncthreads();                                      // the next piece re-uses the LDS stages
  }
}

// the implicit 3x3 weight gradients of up to 16 convolutions in one launch (the same decomposition)
constexpr int kMaxConvJobs = 16;
struct DwpConvJob {
  DwpJob j;
  DwpConv cv;
};
struct DwpConvTable {
  int njobs, work, quota;
  DwpConvJob job[kMaxConvJobs];
};
template <bool SYM>
__global__ __launch_bounds__(512, 1) void dwp_conv_kernel(const DwpConvTable tab) {
  const int id = xcd_contiguous(blockIdx.x, gridDim.x);
  int w0 = id * tab.quota;
  const int w1 = min(tab.work, w0 + tab.quota);
  while (w0 < w1) {
    int lo = 0, hi = tab.njobs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tab.job[mid].j.first_work <= w0) lo = mid; else hi = mid - 1;
    }
    const DwpJob& j = tab.job[lo].j;
    const int local = w0 - j.first_work;
    const int tile = local / j.steps, s = local - tile * j.steps;
    const int n = min(j.steps - s, w1 - w0);
    dwp_body<SYM, 0, true>(j.dY, j.X, j.dW, j.M, j.K, j.L, j.k_tiles, tile, s, n, tab.job[lo].cv);
    w0 += n;
    __syncthreads();
  }
}

// Xs[8 + i] = X[i + 1] for i = -1 .. n - 2, zeros elsewhere (Xs holds n + 16 elements): the copy that makes the horizontal taps
// dword-aligned sources.  The 8-element front pad keeps the chunk that starts one or two elements BEFORE the tensor (kx = 0 at the
// first pixels of the first plane) inside the buffer -- the hardware's range check drops a copy whose first byte is out of range as a
// whole, valid elements included.  8 elements per thread.
__global__ __launch_bounds__(256) void shift1_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ xs,
                                                          int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= n8; i += (int64_t)gridDim.x * 256) {
    if (i == n8) {          // the pads: X[0] is the last element of the front pad, the tail is zero
      const unsigned x0 = x[0];
      *reinterpret_cast<u32x4*>(xs) = u32x4{0u, 0u, 0u, x0 << 16};
      *reinterpret_cast<u32x4*>(xs + 8 + 8 * n8) = u32x4{0u, 0u, 0u, 0u};
      continue;
    }
    const u32x4 a = *reinterpret_cast<const u32x4*>(x + 8 * i);
    const unsigned nxt = i + 1 < n8 ? *reinterpret_cast<const unsigned*>(x + 8 * i + 8) : 0u;
    u32x4 o;
    o.x = (a.x >> 16) | (a.y << 16);
    o.y = (a.y >> 16) | (a.z << 16);
    o.z = (a.z >> 16) | (a.w << 16);
    o.w = (a.w >> 16) | (nxt << 16);
    *reinterpret_cast<u32x4*>(xs + 8 + 8 * i) = o;
  }
}

bool dwp_shape_ok(int batch, int M, int K, int L) {
  if (!(batch > 0 && M > 0 && K > 0 && L >= 32 && (L & 3) == 0 && (int64_t)batch * ((L + 31) >> 5) < (1ll << 30))) return false;
  if (L & 31) return (int64_t)batch * M * L < (1ll << 30) && (int64_t)batch * K * L < (1ll << 31);          // ragged: whole-tensor offsets
  return (int64_t)M * L < (1ll << 30) && (int64_t)K * L < (1ll << 31);
}

}  // namespace

// 1 when the pipelined kernel takes this shape (the host's dispatch asks before choosing it)
extern "C" int s2f_spike_gemm_dw_pipe_ok(int batch, int M, int K, int L) { return dwp_shape_ok(batch, M, K, L) ? 1 : 0; }

static int dwp_launch_table(DwpJobTable& tab, int cfg, int target_wgs, void* stream, const char* who) {
  int64_t work = 0;
  for (int i = 0; i < tab.njobs; ++i) {
    DwpJob& j = tab.job[i];
    j.k_tiles = (j.K + TK - 1) / TK;
    j.first_work = (int)work;
    work += (int64_t)((j.M + TM - 1) / TM) * j.k_tiles * j.steps;
    S2F_REQUIRE(work < (1ll << 30), S2F_EINVAL, "%s: too much work for one launch", who);
  }
  // one workgroup per CU (256 on MI355X); at least 8 steps per workgroup
  if (target_wgs <= 0) target_wgs = 256;
  int quota = (int)((work + target_wgs - 1) / target_wgs);
  if (quota < 8) quota = 8;
  const int wgs = (int)((work + quota - 1) / quota);
  tab.work = (int)work, tab.quota = quota;
  hipStream_t s = (hipStream_t)stream;
#ifdef S2F_DWP_PROBE
#define S2F_KO(N) if (cfg == 2 * N + 1) { S2F_LAUNCH(true, true, (dwp_grouped_kernel<true, N>), dim3((unsigned)wgs), dim3(512), 0, s, tab); return s2f_check_launch("ko"); } \
                  if (cfg == 2 * N) { S2F_LAUNCH(true, true, (dwp_grouped_kernel<false, N>), dim3((unsigned)wgs), dim3(512), 0, s, tab); return s2f_check_launch("ko"); }
  S2F_KO(1) S2F_KO(3) S2F_KO(4) S2F_KO(12) S2F_KO(15) S2F_KO(20) S2F_KO(36) S2F_KO(52) S2F_KO(48)
#undef S2F_KO
#endif
  bool ragged = false;
  for (int i = 0; i < tab.njobs; ++i) ragged = ragged || (tab.job[i].L & 31) != 0;
  S2F_REQUIRE(!ragged || cfg == 0, S2F_EINVAL, "%s: L %% 32 != 0 runs on the two-halves schedule only (cfg 0)", who);
  if (cfg == 1)
    S2F_LAUNCH(true, true, (dwp_grouped_kernel<true>), dim3((unsigned)wgs), dim3(512), 0, s, tab);
  else if (ragged)
    S2F_LAUNCH(true, true, (dwp_grouped_kernel<false, 0, true>), dim3((unsigned)wgs), dim3(512), 0, s, tab);
  else
    S2F_LAUNCH(true, true, (dwp_grouped_kernel<false>), dim3((unsigned)wgs), dim3(512), 0, s, tab);
  return s2f_check_launch(who);
}

// dW (+)= sum_b dY[b] X[b]^T on the pipelined kernel.  cfg: 0 = two-halves schedule, 1 = symmetric schedule;
// target_wgs <= 0: one workgroup per CU.
extern "C" int s2f_spike_gemm_dw_pipe(const float* dY, const uint16_t* X, float* dW, int batch, int M, int K, int L,
                                      int accumulate, int cfg, int target_wgs, void* stream) {
  S2F_REQUIRE(dY && X && dW, S2F_EINVAL, "s2f_spike_gemm_dw_pipe: null pointer");
  S2F_REQUIRE(dwp_shape_ok(batch, M, K, L), S2F_EINVAL, "s2f_spike_gemm_dw_pipe: needs L %% 4 == 0, L >= 32, M L < 2^30, K L < 2^31 (M=%d K=%d L=%d)",
              M, K, L);
  S2F_REQUIRE(s2f_aligned16(dY) && s2f_aligned16(X), S2F_EALIGN, "s2f_spike_gemm_dw_pipe: operands must be 16-byte aligned");
  if (!accumulate && s2f_zero_async(dW, sizeof(float) * (size_t)M * K, (hipStream_t)stream) != S2F_OK)
    return s2f_check_launch("s2f_spike_gemm_dw_pipe memset");
  DwpJobTable tab;
  tab.njobs = 1;
  tab.job[0] = DwpJob{dY, X, dW, M, K, L, batch * ((L + 31) >> 5), 0, 0};
  return dwp_launch_table(tab, cfg, target_wgs, stream, "s2f_spike_gemm_dw_pipe");
}

// MANY weight gradients in one launch (what s2f_spike_gemm_dw_grouped is to the round-2 kernel).  jobs (HOST array):
// njobs x {dY, X, dW (pointers), batch, M, K, L}; every dW is accumulated into.
extern "C" int s2f_spike_gemm_dw_pipe_grouped(const int64_t* jobs, int njobs, int cfg, int target_wgs, void* stream) {
  S2F_REQUIRE(jobs && njobs > 0 && njobs <= kMaxJobs, S2F_EINVAL, "s2f_spike_gemm_dw_pipe_grouped: 1 .. %d jobs", kMaxJobs);
  DwpJobTable tab;
  tab.njobs = njobs;
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
    j.steps = B * ((j.L + 31) >> 5);
  }
  return dwp_launch_table(tab, cfg, target_wgs, stream, "s2f_spike_gemm_dw_pipe_grouped");
}

// 1 when the pipelined kernel takes the implicit 3x3 weight gradient of this shape
extern "C" int s2f_spike_conv3x3_dw_pipe_ok(int batch, int M, int C, int H, int W) {
  return (C > 0 && C % 32 == 0 && H > 0 && W >= 32 && W % 8 == 0 && (H * W) % 32 == 0 && dwp_shape_ok(batch, M, 9 * C, H * W) &&
          (int64_t)batch * C * H * W * 2 < (1ll << 31)) ? 1 : 0;
}

// Xs (n + 16 elements): Xs[8 + i] = X[i + 1], i = -1 .. n - 2, zeros elsewhere (n % 8 == 0; both 16-byte aligned): the shifted copy
// s2f_spike_conv3x3_dw_pipe reads the horizontal taps from
extern "C" int s2f_shift1_bf16(const uint16_t* X, uint16_t* Xs, int64_t n, void* stream) {
  S2F_REQUIRE(X && Xs && n > 0 && (n & 7) == 0, S2F_EINVAL, "s2f_shift1_bf16: null pointer or n %% 8 != 0");
  S2F_REQUIRE(s2f_aligned16(X) && s2f_aligned16(Xs), S2F_EALIGN, "s2f_shift1_bf16: 16-byte alignment");
  int64_t blocks = (n / 8 + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks + 1;          // (+1: the thread that writes the pads)
  hipLaunchKernelGGL(shift1_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, Xs, n / 8);
  return s2f_check_launch("s2f_shift1_bf16");
}

// Implicit 3x3 weight gradients (stride 1, padding 1) on the pipelined kernel, many convolutions per launch: jobs (HOST array):
// njobs x {dY, X, Xs, dW (pointers), batch, M, C, H, W}; dW [M][3][3][C] tap-major (as s2f_spike_conv3x3_dw_bf16) or -- cfg & 2 --
// [M][C][3][3], the weight's own layout (straight into the parameter's gradient slot), accumulated into.  Xs = s2f_shift1_bf16(X).  Replaces the autograd weight gradient of MS_ConvBlock's dense 3x3 convolutions
// (mmseg/models/backbones/sdtv2.py:183-219) and of the stride-1 down-sampling (sdtv2.py:540-548).
extern "C" int s2f_spike_conv3x3_dw_pipe(const int64_t* jobs, int njobs, int cfg, int target_wgs, void* stream) {
  S2F_REQUIRE(jobs && njobs > 0 && njobs <= kMaxConvJobs, S2F_EINVAL, "s2f_spike_conv3x3_dw_pipe: 1 .. %d jobs", kMaxConvJobs);
  DwpConvTable tab;
  tab.njobs = njobs;
  int64_t work = 0;
  for (int i = 0; i < njobs; ++i) {
    const int64_t* r = jobs + 9 * i;
    DwpConvJob& cj = tab.job[i];
    cj.j.dY = reinterpret_cast<const float*>(r[0]);
    cj.j.X = reinterpret_cast<const unsigned short*>(r[1]);
    cj.cv.Xs = reinterpret_cast<const unsigned short*>(r[2]);
    cj.j.dW = reinterpret_cast<float*>(r[3]);
    const int B = (int)r[4], C = (int)r[6], H = (int)r[7], W = (int)r[8];
    cj.j.M = (int)r[5];
    S2F_REQUIRE(cj.j.dY && cj.j.X && cj.cv.Xs && cj.j.dW && s2f_spike_conv3x3_dw_pipe_ok(B, cj.j.M, C, H, W), S2F_EINVAL,
                "s2f_spike_conv3x3_dw_pipe: bad job %d (needs C %% 32 == 0, W %% 8 == 0, W >= 32, H W %% 32 == 0)", i);
    S2F_REQUIRE(s2f_aligned16(cj.j.dY) && s2f_aligned16(cj.j.X) && s2f_aligned16(cj.cv.Xs), S2F_EALIGN,
                "s2f_spike_conv3x3_dw_pipe: job %d misaligned", i);
    cj.j.K = 9 * C, cj.j.L = H * W, cj.j.steps = B * (cj.j.L >> 5);
    cj.cv.C = C, cj.cv.H = H, cj.cv.W = W, cj.cv.x_bytes = (unsigned)((int64_t)B * C * H * W * 2);
    cj.cv.wl = (cfg & 2) ? 1 : 0;
    cj.j.k_tiles = (cj.j.K + TK - 1) / TK;
    cj.j.first_work = (int)work;
    work += (int64_t)((cj.j.M + TM - 1) / TM) * cj.j.k_tiles * cj.j.steps;
    S2F_REQUIRE(work < (1ll << 30), S2F_EINVAL, "s2f_spike_conv3x3_dw_pipe: too much work for one launch");
  }
  if (target_wgs <= 0) target_wgs = 256;
  int quota = (int)((work + target_wgs - 1) / target_wgs);
  if (quota < 8) quota = 8;
  const int wgs = (int)((work + quota - 1) / quota);
  tab.work = (int)work, tab.quota = quota;
  hipStream_t s = (hipStream_t)stream;
  if (cfg & 1)
    S2F_LAUNCH(true, true, (dwp_conv_kernel<true>), dim3((unsigned)wgs), dim3(512), 0, s, tab);
  else
    S2F_LAUNCH(true, true, (dwp_conv_kernel<false>), dim3((unsigned)wgs), dim3(512), 0, s, tab);
  return s2f_check_launch("s2f_spike_conv3x3_dw_pipe");
}
