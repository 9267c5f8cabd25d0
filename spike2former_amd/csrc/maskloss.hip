// Hungarian-matched MaskFormer loss against SEMANTIC maps, for gfx950.
//
// mmseg's MaskFormerHead turns each semantic map into one binary mask per class present (_seg_data_to_instance_data,
// mmseg/models/decode_heads/maskformer_head.py:53-106), so the targets of one image are DISJOINT and every one of them is
// "seg == class".  The reference then (a) forms three [L*Q, h*w] x [h*w, n_gt] products per image for the matching costs
// (match_cost.py:289-297 FocalLossCost(binary_input), :361-371 DiceCost) after ~10 element-wise passes over the logits, and
// (b) gathers the matched logits and targets into [num_masks, H, W] tensors for the mask losses
// (dense_heads/maskformer_head.py:462-494).  With disjoint targets neither is needed:
//   * the three products against 0/1 columns are SEGMENTED SUMS of per-pixel terms by the pixel's label: one pass over the logits,
//     bins for every class id (the host keeps the columns of the classes present) -- mask_cost_bins_kernel;
//   * the target of a matched (image, layer, query) row is a class id: the loss kernels compare the label map on the fly, rows
//     without a match are skipped; nothing is gathered, no [num_masks, H, W] tensor exists forward or backward (the adjoint of the
//     2x bilinear up-sampling is applied to an LDS tile of the per-pixel derivatives) -- mask_loss_seg_{fwd,bwd}_kernel.
// All shapes are independent of the matching, so the whole loss replays inside a hipGraph (graph.GraphedHungarianStep).
//
// Repeatability.  The matching costs are 64-bit fixed-point sums (integer adds: the assignment is identical from run to run).  The
// four loss sums of a row (mask_loss_seg_fwd_kernel) are bit-repeatable since round 5: a row is cut into `chunks` (32 at C2) pieces
// whose partial sums are STORED to partials[row][chunk][4] and added in chunk order by mask_loss_seg_finalize_kernel (which also
// writes the zeros of the rows without a match: the clearing launch it replaces) -- rounds 2-4 added them to sums[row] with fp32
// atomics in arrival order.  (The layer-scale gradient of transpose_scale_add_bwd_kernel, transpose.hip, still uses atomics.)
#include "s2f_common.h"

#pragma clang fp contract(off)

namespace {

// torch computes  lambda = src - floor(src) and  (1-lambda)*a + lambda*b  (align_corners = False, scale 2)
__device__ __forceinline__ void taps2(int o, int in_size, int& i0, int& i1, float& l1) {
  float src = ((float)o + 0.5f) * 0.5f - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

struct Sig {
  float s, c, l1p;                  // sigmoid(u), 1 - sigmoid(u) (without cancellation), log(1 + exp(-|u|))
};
__device__ __forceinline__ Sig sigmoid_parts(float u) {
  const float e = __expf(-fabsf(u));
  const float inv = __builtin_amdgcn_rcpf(1.f + e);
  const float small = e * inv;
  Sig r;
  r.s = u >= 0.f ? inv : small;
  r.c = u >= 0.f ? small : inv;
  r.l1p = __logf(1.f + e);
  return r;
}
__device__ __forceinline__ float pw(float x, float gamma) { return gamma == 2.f ? x * x : __powf(x, gamma); }

// ---------------------------------------------------------------------------------------------------------------------
// Matching costs.  For image b, prediction row r (= layer * Q + query) and every class id c < K:
//   out[b][r][c]       = sum_{pixels of class c} (pos - neg)          pos = -log(s + eps) alpha (1 - s)^gamma
//   out[b][r][K + c]   = sum_{pixels of class c} s                    neg = -log(1 - s + eps) (1 - alpha) s^gamma
//   out[b][r][2K]      = sum_{all pixels} neg,   out[b][r][2K + 1] = sum_{all pixels} s
// so that  pos @ g^T + neg @ (1 - g)^T = out[c] + out[2K]  and  s @ g^T = out[K + c]  for the 0/1 column g of class c.
// Sums are accumulated as 64-bit fixed point (2^-32): integer addition is associative, the result does not depend on the order
// the LDS atomics retire in -- the assignment must not change from run to run.  A thread keeps a running (label, sums) pair and
// touches LDS only when the label changes: label maps are piecewise constant.
constexpr float kFix = 4294967296.f;
constexpr int kMaxClasses = 256;

__global__ __launch_bounds__(256) void mask_cost_bins_kernel(const float* __restrict__ pred, const unsigned char* __restrict__ seg,
                                                             float* __restrict__ out, int R, int hw, int K, float alpha,
                                                             float gamma, float eps) {
  __shared__ unsigned long long bins[2 * kMaxClasses + 2];
  const int r = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* p = pred + ((int64_t)b * R + r) * hw;
  const unsigned char* sg = seg + (int64_t)b * hw;
  for (int i = tid; i < 2 * K + 2; i += 256) bins[i] = 0ull;
  __syncthreads();
  long long negtot = 0, stot = 0, accD = 0, accS = 0;
  int cur = -1;
  auto flush = [&]() {
    if (cur >= 0 && cur < K) {
      atomicAdd(&bins[cur], (unsigned long long)accD);
      atomicAdd(&bins[K + cur], (unsigned long long)accS);
    }
  };
  for (int q = tid; q < hw / 4; q += 256) {
    const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
    const uchar4 lv = *reinterpret_cast<const uchar4*>(sg + 4 * q);
    const float u[4] = {v.x, v.y, v.z, v.w};
    const int lab[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const Sig g = sigmoid_parts(u[j]);
      const float pos = -__logf(g.s + eps) * alpha * pw(g.c, gamma);
      const float neg = -__logf(g.c + eps) * (1.f - alpha) * pw(g.s, gamma);
      const long long dp = __float2ll_rn(pos * kFix), dn = __float2ll_rn(neg * kFix), ds = __float2ll_rn(g.s * kFix);
      negtot += dn;
      stot += ds;
      if (lab[j] != cur) {
        flush();
        cur = lab[j];
        accD = accS = 0;
      }
      accD += dp - dn;
      accS += ds;
    }
  }
  flush();
  for (int o = 32; o > 0; o >>= 1) {
    negtot += __shfl_xor(negtot, o, 64);
    stot += __shfl_xor(stot, o, 64);
  }
  if ((tid & 63) == 0) {
    atomicAdd(&bins[2 * K], (unsigned long long)negtot);
    atomicAdd(&bins[2 * K + 1], (unsigned long long)stot);
  }
  __syncthreads();
  float* o = out + ((int64_t)b * R + r) * (2 * K + 2);
  for (int i = tid; i < 2 * K + 2; i += 256) o[i] = (float)((double)(long long)bins[i] * (1.0 / 4294967296.0));
}

// ---------------------------------------------------------------------------------------------------------------------
// Mask losses.  Row = (image b, prediction r); its target is  seg[b] == row_class[row]  (row_class < 0: no match, skipped).
//   u = bilinear2x(pred[row]),  s = sigmoid(u),  sums[row] = { sum s t, sum s, sum t, sum focal(u, t) }
//   focal(u, t) = BCEWithLogits(u, t) (alpha t + (1 - alpha)(1 - t)) ((1 - s) t + s (1 - t))^gamma   (losses/focal_loss.py:36-44)
struct Pix {
  float s, t, bce, pt, at;
};
__device__ __forceinline__ Pix pix(float u, bool hit, float alpha) {
  const Sig g = sigmoid_parts(u);
  Pix m;
  m.t = hit ? 1.f : 0.f;
  m.s = g.s;
  m.bce = fmaxf(u, 0.f) - u * m.t + g.l1p;
  m.pt = hit ? g.c : g.s;
  m.at = hit ? alpha : 1.f - alpha;
  return m;
}

// 4 consecutive up-sampled values of one output row from the two source rows r0 / r1 (vertical weight ly on r1)
__device__ __forceinline__ void up4(const float* __restrict__ r0, const float* __restrict__ r1, float ly, int q, int w, float (&u)[4]) {
  const int m = 2 * q;                                   // source columns m-1, m, m+1, m+2 (clamped)
  const int jm = m > 0 ? m - 1 : 0, jp = m + 2 < w ? m + 2 : w - 1;
  const float2 a = *reinterpret_cast<const float2*>(r0 + m), c = *reinterpret_cast<const float2*>(r1 + m);
  const float am = r0[jm], ap = r0[jp], cm = r1[jm], cp = r1[jp];
  const float top[4] = {0.25f * am + 0.75f * a.x, 0.75f * a.x + 0.25f * a.y, 0.25f * a.x + 0.75f * a.y, 0.75f * a.y + 0.25f * ap};
  const float bot[4] = {0.25f * cm + 0.75f * c.x, 0.75f * c.x + 0.25f * c.y, 0.25f * c.x + 0.75f * c.y, 0.75f * c.y + 0.25f * cp};
#pragma unroll
  for (int j = 0; j < 4; ++j) u[j] = (1.f - ly) * top[j] + ly * bot[j];
}

__global__ __launch_bounds__(256) void mask_loss_seg_fwd_kernel(const float* __restrict__ pred, const unsigned char* __restrict__ seg,
                                                                const int* __restrict__ row_class, float* __restrict__ part, int R,
                                                                int h, int w, float alpha, float gamma, int chunks) {
  const int row = blockIdx.y;
  const int cls = row_class[row];
  if (cls < 0) return;                                   // (the finalize kernel writes this row's zeros)
  const int W = 2 * w, H = 2 * h;
  const float* pp = pred + (int64_t)row * h * w;
  const unsigned char* sp = seg + (int64_t)(row / R) * H * W;
  const int quads = W / 4;
  const int total = H * quads;
  const int per = (total + chunks - 1) / chunks;
  const int beg = blockIdx.x * per, end = min(beg + per, total);
  float a = 0.f, b = 0.f, c = 0.f, f = 0.f;
  for (int idx = beg + threadIdx.x; idx < end; idx += 256) {
    const int q = idx % quads, oy = idx / quads;
    int y0, y1;
    float ly;
    taps2(oy, h, y0, y1, ly);
    float u[4];
    up4(pp + (int64_t)y0 * w, pp + (int64_t)y1 * w, ly, q, w, u);
    const uchar4 tv = *reinterpret_cast<const uchar4*>(sp + (int64_t)oy * W + q * 4);
    const int lab[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const Pix m = pix(u[j], lab[j] == cls, alpha);
      a += m.s * m.t;
      b += m.s;
      c += m.t;
      f += m.bce * m.at * pw(m.pt, gamma);
    }
  }
  __shared__ float red[4][4];
  float v[4] = {a, b, c, f};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    if (lane == 0) red[wave][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 4)
    part[((int64_t)row * chunks + blockIdx.x) * 4 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// sums[row][k] = partials of the row's chunks added in chunk order (rows without a match: zero)
__global__ __launch_bounds__(256) void mask_loss_seg_finalize_kernel(const float* __restrict__ part, const int* __restrict__ row_class,
                                                                     float* __restrict__ sums, int rows, int chunks) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * 4) return;
  const int row = i >> 2, k = i & 3;
  float a = 0.f;
  if (row_class[row] >= 0)
    for (int c = 0; c < chunks; ++c) a += part[((int64_t)row * chunks + c) * 4 + k];
  sums[i] = a;
}

// Backward: gpred[row] = adjoint of the 2x up-sampling applied to  d(sum_k g[row][k] sums[row][k]) / du.  A workgroup owns a
// TY x TX tile of the low-resolution gradient: it evaluates the derivative on the (2 TY + 2) x (2 TX + 2) up-sampled pixels that
// touch the tile (one halo pixel per side: 14 % more evaluations than pixels) into LDS, then every low-resolution pixel gathers
// its 4 x 4 footprint with the fixed weights {0.25, 0.75, 0.75, 0.25} (the border taps fold back: weight 1 on the first / last
// up-sampled row and column).  Rows without a match get zeros.
constexpr int TY = 8, TX = 64;
constexpr int LY = TY + 2, LX = TX + 2;                  // staged low-resolution logits (one halo pixel)
constexpr int DY = 2 * TY + 2, DX = 2 * TX + 2, DXP = DX + 2;

__global__ __launch_bounds__(256) void mask_loss_seg_bwd_kernel(const float* __restrict__ pred, const unsigned char* __restrict__ seg,
                                                                const int* __restrict__ row_class, const float* __restrict__ g,
                                                                float* __restrict__ gpred, int R, int h, int w, float alpha,
                                                                float gamma, int tiles_x) {
  __shared__ float Ls[LY][LX + 1];
  __shared__ float Ds[DY][DXP];
  const int row = blockIdx.y, tid = threadIdx.x;
  const int iy0 = (blockIdx.x / tiles_x) * TY, ix0 = (blockIdx.x % tiles_x) * TX;
  float* gp = gpred + (int64_t)row * h * w;
  const int cls = row_class[row];
  if (cls < 0) {
    for (int c = tid; c < TY * TX; c += 256) {
      const int i = iy0 + c / TX, j = ix0 + c % TX;
      if (i < h && j < w) gp[(int64_t)i * w + j] = 0.f;
    }
    return;
  }
  const int W = 2 * w, H = 2 * h;
  const float* pp = pred + (int64_t)row * h * w;
  const unsigned char* sp = seg + (int64_t)(row / R) * H * W;
  const float ga = g[row * 4], gb = g[row * 4 + 1], gf = g[row * 4 + 3];
  for (int c = tid; c < LY * LX; c += 256) {
    const int li = c / LX, lj = c % LX;
    const int i = min(max(iy0 - 1 + li, 0), h - 1), j = min(max(ix0 - 1 + lj, 0), w - 1);
    Ls[li][lj] = pp[(int64_t)i * w + j];
  }
  __syncthreads();
  const int oy0 = 2 * iy0 - 1, ox0 = 2 * ix0 - 1;
  for (int c = tid; c < DY * DX; c += 256) {
    const int dy = c / DX, dx = c % DX;
    const int oy = oy0 + dy, ox = ox0 + dx;
    float d = 0.f;
    if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
      int y0, y1, x0, x1;
      float ly, lx;
      taps2(oy, h, y0, y1, ly);
      taps2(ox, w, x0, x1, lx);
      const int ly0 = y0 - (iy0 - 1), ly1 = y1 - (iy0 - 1), lx0 = x0 - (ix0 - 1), lx1 = x1 - (ix0 - 1);
      const float top = (1.f - lx) * Ls[ly0][lx0] + lx * Ls[ly0][lx1];
      const float bot = (1.f - lx) * Ls[ly1][lx0] + lx * Ls[ly1][lx1];
      const float u = (1.f - ly) * top + ly * bot;
      const Pix m = pix(u, sp[(int64_t)oy * W + ox] == cls, alpha);
      const float ds = m.s * (1.f - m.s);                                   // ds/du
      const float dpt = ds * (1.f - 2.f * m.t);                             // d pt / du
      const float ptg1 = gamma == 2.f ? m.pt : __powf(m.pt, gamma - 1.f);   // pt^(gamma-1)
      const float dfocal = m.at * ((m.s - m.t) * ptg1 * m.pt + m.bce * gamma * ptg1 * dpt);
      d = (ga * m.t + gb) * ds + gf * dfocal;
    }
    Ds[dy][dx] = d;
  }
  __syncthreads();
  for (int c = tid; c < TY * TX; c += 256) {
    const int ti = c / TX, tj = c % TX;
    const int i = iy0 + ti, j = ix0 + tj;
    if (i >= h || j >= w) continue;
    // up-sampled rows 2i-1 .. 2i+2 sit at Ds rows 2 ti .. 2 ti + 3 (columns alike); pixels outside the image hold zeros
    const float wy[4] = {0.25f, i == 0 ? 1.f : 0.75f, i == h - 1 ? 1.f : 0.75f, 0.25f};
    const float wx[4] = {0.25f, j == 0 ? 1.f : 0.75f, j == w - 1 ? 1.f : 0.75f, 0.25f};
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float* dr = &Ds[2 * ti + a][2 * tj];
      acc += wy[a] * ((wx[0] * dr[0] + wx[1] * dr[1]) + (wx[2] * dr[2] + wx[3] * dr[3]));
    }
    gp[(int64_t)i * w + j] = acc;
  }
}

}  // namespace

extern "C" int s2f_mask_cost_bins(const float* pred, const uint8_t* seg_small, float* out, int B, int R, int64_t hw, int K,
                                  float alpha, float gamma, float eps, void* stream) {
  S2F_REQUIRE(pred && seg_small && out, S2F_EINVAL, "s2f_mask_cost_bins: null pointer");
  S2F_REQUIRE(B > 0 && B < 65536 && R > 0 && hw > 0 && hw < (1ll << 31) && (hw & 3) == 0 && K > 0 && K <= kMaxClasses, S2F_EINVAL,
              "s2f_mask_cost_bins: need hw %% 4 == 0, 0 < K <= 256 (hw=%lld, K=%d)", (long long)hw, K);
  S2F_REQUIRE(s2f_aligned16(pred) && (reinterpret_cast<uintptr_t>(seg_small) & 3u) == 0, S2F_EALIGN,
              "s2f_mask_cost_bins: logits 16-byte, label map 4-byte aligned");
  hipLaunchKernelGGL(mask_cost_bins_kernel, dim3((unsigned)R, (unsigned)B), dim3(256), 0, (hipStream_t)stream, pred, seg_small, out,
                     R, (int)hw, K, alpha, gamma, eps);
  return s2f_check_launch("s2f_mask_cost_bins");
}

static int seg_chunks(int h, int w) {
  const int64_t total = (int64_t)2 * h * (2 * w / 4);
  const int chunks = (int)((total + 256 * 8 - 1) / (256 * 8));       // >= 8 iterations per thread
  return chunks < 1 ? 1 : chunks;
}

extern "C" int64_t s2f_mask_loss_seg_partials(int B, int R, int h, int w) { return (int64_t)B * R * seg_chunks(h, w) * 4; }

extern "C" int s2f_mask_loss_seg_fwd(const float* pred, const uint8_t* seg, const int32_t* row_class, float* sums, float* partials,
                                     int B, int R, int h, int w, float alpha, float gamma, void* stream) {
  S2F_REQUIRE(pred && seg && row_class && sums && partials, S2F_EINVAL, "s2f_mask_loss_seg_fwd: null pointer");
  S2F_REQUIRE(B > 0 && R > 0 && (int64_t)B * R < 65536 && h > 0 && w > 0 && (w % 2) == 0, S2F_EINVAL,
              "s2f_mask_loss_seg_fwd: need B * R < 65536 and an even width");
  S2F_REQUIRE((reinterpret_cast<uintptr_t>(seg) & 3u) == 0 && (reinterpret_cast<uintptr_t>(pred) & 7u) == 0, S2F_EALIGN,
              "s2f_mask_loss_seg_fwd: label map 4-byte, logits 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int chunks = seg_chunks(h, w);
  hipLaunchKernelGGL(mask_loss_seg_fwd_kernel, dim3(chunks, (unsigned)(B * R)), dim3(256), 0, s, pred, seg, row_class, partials, R, h,
                     w, alpha, gamma, chunks);
  hipLaunchKernelGGL(mask_loss_seg_finalize_kernel, dim3((unsigned)((B * R * 4 + 255) / 256)), dim3(256), 0, s, partials, row_class,
                     sums, B * R, chunks);
  return s2f_check_launch("s2f_mask_loss_seg_fwd");
}

extern "C" int s2f_mask_loss_seg_bwd(const float* pred, const uint8_t* seg, const int32_t* row_class, const float* g_sums,
                                     float* gpred, int B, int R, int h, int w, float alpha, float gamma, void* stream) {
  S2F_REQUIRE(pred && seg && row_class && g_sums && gpred, S2F_EINVAL, "s2f_mask_loss_seg_bwd: null pointer");
  S2F_REQUIRE(B > 0 && R > 0 && (int64_t)B * R < 65536 && h > 0 && w > 0 && (w % 2) == 0, S2F_EINVAL,
              "s2f_mask_loss_seg_bwd: need B * R < 65536 and an even width");
  const int tiles_x = (w + TX - 1) / TX, tiles_y = (h + TY - 1) / TY;
  hipLaunchKernelGGL(mask_loss_seg_bwd_kernel, dim3(tiles_x * tiles_y, (unsigned)(B * R)), dim3(256), 0, (hipStream_t)stream, pred, seg,
                     row_class, g_sums, gpred, R, h, w, alpha, gamma, tiles_x);
  return s2f_check_launch("s2f_mask_loss_seg_bwd");
}
