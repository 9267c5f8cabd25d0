// Exact-2x bilinear up-sampling (align_corners = False) of [N*C, h, w] planes and its adjoint, for gfx950.
//
// Reference call site: the FPN top-down path of the pixel decoder,
//   y = cur_feat + F.interpolate(y, size=cur_feat.shape[-2:], mode='bilinear', align_corners=False)
// (mmdet/models/layers/pixel_decoder.py:456-460); every level doubles the resolution (H/16 -> H/8 -> H/4 -> H/2).
// ATen's generic kernel needs 0.80 ms for the largest level ([8,256,256,256]); this is a pure HBM stream:
// with scale 2 the source index of output o is  o/2 - 0.25  clamped at 0, i.e. fixed weights
//   out[2i]   = 0.25 * in[max(i-1,0)] + 0.75 * in[i]        out[2i+1] = 0.75 * in[i] + 0.25 * in[min(i+1,w-1)]
// in each dimension.  One thread produces 4 consecutive outputs of a row (one 16-byte store) from a 2 x 4 input patch.
#include "s2f_common.h"

#pragma clang fp contract(off)

namespace {

// torch computes  lambda = src - floor(src) and  (1-lambda)*a + lambda*b ; reproduce that association exactly.
__device__ __forceinline__ void taps(int o, int in_size, int& i0, int& i1, float& l1) {
  float src = ((float)o + 0.5f) * 0.5f - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

__global__ __launch_bounds__(256) void up2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t planes,
                                                       int h, int w) {
  const int W = 2 * w, H = 2 * h;
  const int64_t quads_per_row = W / 4;
  const int64_t total = planes * H * quads_per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx % quads_per_row);
    const int64_t r = idx / quads_per_row;
    const int oy = (int)(r % H);
    const int64_t p = r / H;
    int y0, y1;
    float ly;
    taps(oy, h, y0, y1, ly);
    const float* r0 = x + (p * h + y0) * w;
    const float* r1 = x + (p * h + y1) * w;
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x0, x1;
      float lx;
      taps(q * 4 + j, w, x0, x1, lx);
      const float top = (1.f - lx) * r0[x0] + lx * r0[x1];
      const float bot = (1.f - lx) * r1[x0] + lx * r1[x1];
      out[j] = (1.f - ly) * top + ly * bot;
    }
    *reinterpret_cast<float4*>(y + (p * H + oy) * W + q * 4) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

// thread -> (plane within the workgroup's group, cell of the plane): planes with fewer than 256 cells are packed 256/cells
// to a workgroup (ppb), larger planes take gridDim.x workgroups each
struct PlaneCell {
  uint32_t cell, sub, ppb;
};
__device__ __forceinline__ PlaneCell plane_cell(uint32_t cells) {
  PlaneCell r;
  if (cells >= 256u) {
    r.cell = blockIdx.x * blockDim.x + threadIdx.x;
    r.sub = 0;
    r.ppb = 1;
  } else {
    r.ppb = 256u / cells;
    r.sub = threadIdx.x / cells;
    r.cell = r.sub < r.ppb ? threadIdx.x - r.sub * cells : cells;      // surplus threads idle
  }
  return r;
}
inline dim3 plane_grid(int64_t cells, int64_t planes) {
  const int64_t ppb = cells >= 256 ? 1 : 256 / cells;
  int64_t gy = (planes + ppb - 1) / ppb;
  if (gy > 65535) gy = 65535;
  return dim3((unsigned)((cells + 255) / 256), (unsigned)gy);
}

// w % 4 == 0: one thread produces a 2 x 8 block of outputs (rows 2i, 2i+1; columns 8k .. 8k+7) from input rows i-1, i, i+1
// and columns 4k-1 .. 4k+4 -- one 16-byte load and two edge scalars per row (9 load instructions for 16 outputs instead of 32
// scalar loads), same tap arithmetic as the kernel above ((1 - l) a + l b with l = 0.75 / 0.25 / 0 at the clamped edge).
// SIGMOID: y = sigmoid(up(x)) -- the inference post-processing's `mask_pred.sigmoid()` (mmseg decode_heads/maskformer_head.py:176)
// inside the up-sampling pass: the 4x larger map is written once instead of written, read and written again.
template <bool SIGMOID>
__global__ __launch_bounds__(256) void up2x_fwd_block_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t planes,
                                                             int h, int w) {
  // grid (ceil(h * w/4 / 256), planes): 32-bit index arithmetic only -- with one flat 64-bit index the four 64-bit
  // divisions per thread (~600 VALU instructions for 16 outputs) held this HBM stream at 0.9 TB/s
  const int W = 2 * w;
  const uint32_t qw = (uint32_t)w / 4u;
  const uint32_t cells = (uint32_t)h * qw;
  const PlaneCell pc = plane_cell(cells);                // small planes: several planes share one workgroup
  if (pc.cell >= cells) return;
  const int i = (int)(pc.cell / qw), k = (int)(pc.cell - (uint32_t)i * qw);
  for (int64_t p = (int64_t)blockIdx.y * pc.ppb + pc.sub; p < planes; p += (int64_t)gridDim.y * pc.ppb) {
    const float* base = x + p * h * w;
    float c[3][6];                                       // rows i-1, i, i+1 (clamped); columns 4k-1 .. 4k+4 (clamped)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const int iy = min(max(i - 1 + rr, 0), h - 1);
      const float* row = base + (int64_t)iy * w + 4 * k;
      const float4 v = *reinterpret_cast<const float4*>(row);
      c[rr][0] = k > 0 ? row[-1] : v.x;
      c[rr][1] = v.x; c[rr][2] = v.y; c[rr][3] = v.z; c[rr][4] = v.w;
      c[rr][5] = k + 1 < (int)qw ? row[4] : v.w;
    }
    // horizontal pass: output column 8k + 2m     = 0.25 c[m] + 0.75 c[m+1]   (first column of the plane: c[1] alone)
    //                  output column 8k + 2m + 1 = 0.75 c[m+1] + 0.25 c[m+2] (last column of the plane: the clamped tap = c[m+1])
    float hx[3][8];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bool first = k == 0 && m == 0;
        hx[rr][2 * m] = first ? (1.f - 0.f) * c[rr][1] + 0.f * c[rr][2] : (1.f - 0.75f) * c[rr][m] + 0.75f * c[rr][m + 1];
        hx[rr][2 * m + 1] = (1.f - 0.25f) * c[rr][m + 1] + 0.25f * c[rr][m + 2];
      }
    float* o = y + (p * 2 * h + 2 * i) * W + 8 * k;
    float ev[8], od[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ev[j] = i == 0 ? (1.f - 0.f) * hx[1][j] + 0.f * hx[2][j] : (1.f - 0.75f) * hx[0][j] + 0.75f * hx[1][j];   // output row 2i
      od[j] = (1.f - 0.25f) * hx[1][j] + 0.25f * hx[2][j];                                                       // output row 2i + 1
      if (SIGMOID) {
        ev[j] = 1.f / (1.f + expf(-ev[j]));
        od[j] = 1.f / (1.f + expf(-od[j]));
      }
    }
    *reinterpret_cast<float4*>(o) = make_float4(ev[0], ev[1], ev[2], ev[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(ev[4], ev[5], ev[6], ev[7]);
    *reinterpret_cast<float4*>(o + W) = make_float4(od[0], od[1], od[2], od[3]);
    *reinterpret_cast<float4*>(o + W + 4) = make_float4(od[4], od[5], od[6], od[7]);
  }
}

// adjoint: gx[i] gathers from the (at most) 4 outputs per dimension that read it
__device__ __forceinline__ float wgt(int o, int i, int in_size) {     // d out[o] / d in[i] along one dimension
  int i0, i1;
  float l1;
  taps(o, in_size, i0, i1, l1);
  float r = 0.f;
  if (i0 == i) r += 1.f - l1;
  if (i1 == i) r += l1;
  return r;
}

// (add?: the gradient a second reader of the up-sampling's INPUT sends back -- s2f_upsample2x_bwd_add -- summed into the result)
__global__ __launch_bounds__(256) void up2x_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int64_t planes,
                                                       int h, int w, const float* __restrict__ add) {
  const int W = 2 * w, H = 2 * h;
  const int64_t total = planes * h * w;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ix = (int)(idx % w);
    const int64_t r = idx / w;
    const int iy = (int)(r % h);
    const int64_t p = r / h;
    float acc = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 2; ++dy) {
      const int oy = 2 * iy + dy;
      if (oy < 0 || oy >= H) continue;
      const float wy = wgt(oy, iy, h);
      if (wy == 0.f) continue;
      const float* row = gy + (p * H + oy) * W;
      float s = 0.f;
#pragma unroll
      for (int dx = -1; dx <= 2; ++dx) {
        const int ox = 2 * ix + dx;
        if (ox < 0 || ox >= W) continue;
        s += wgt(ox, ix, w) * row[ox];
      }
      acc += wy * s;
    }
    gx[idx] = add ? acc + add[idx] : acc;
  }
}

// Adjoint for w % 4 == 0, h % 2 == 0: one thread produces a 2 x 4 block of gx from 6 rows x 10 columns of gy -- two
// 16-byte loads and two edge scalars per row (3 load instructions per output instead of 16), fixed weights
//   gx[i] = 0.25 gy[2i-1] + 0.75 gy[2i] + 0.75 gy[2i+1] + 0.25 gy[2i+2]   per dimension,
// where at the plane's first / last index the missing outer tap's weight moves to the inner one (the clamped source index
// of the forward: out[0] = in[0], out[2w-1] = in[w-1]).
__global__ __launch_bounds__(256) void up2x_bwd_block_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                             int64_t planes, int h, int w, const float* __restrict__ add) {
  const int W = 2 * w, H = 2 * h;
  const int qw = w / 4, qh = h / 2;
  const uint32_t cells = (uint32_t)qh * (uint32_t)qw;                 // grid (ceil(cells / 256), plane groups), as the forward
  const PlaneCell pc = plane_cell(cells);
  if (pc.cell >= cells) return;
  const int jy = (int)(pc.cell / (uint32_t)qw), k = (int)(pc.cell - (uint32_t)jy * (uint32_t)qw);
  for (int64_t p = (int64_t)blockIdx.y * pc.ppb + pc.sub; p < planes; p += (int64_t)gridDim.y * pc.ppb) {
    const int iy = 2 * jy, ix0 = 4 * k;
    const float* base = gy + p * H * W;
    const float wfirst = ix0 == 0 ? 1.f : 0.75f, wlast = ix0 + 4 == w ? 1.f : 0.75f;
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
      const int oy = 2 * iy - 1 + rr;
      if (oy < 0 || oy >= H) continue;
      const float* row = base + (int64_t)oy * W + 8 * k;
      const float4 a = *reinterpret_cast<const float4*>(row), b = *reinterpret_cast<const float4*>(row + 4);
      const float left = k > 0 ? row[-1] : 0.f, right = k + 1 < qw ? row[8] : 0.f;
      float hx[4];
      hx[0] = (0.25f * left + wfirst * a.x) + (0.75f * a.y + 0.25f * a.z);
      hx[1] = (0.25f * a.y + 0.75f * a.z) + (0.75f * a.w + 0.25f * b.x);
      hx[2] = (0.25f * a.w + 0.75f * b.x) + (0.75f * b.y + 0.25f * b.z);
      hx[3] = (0.25f * b.y + 0.75f * b.z) + (wlast * b.w + 0.25f * right);
      // vertical weights of gy row oy on gx rows iy (taps rr = 0..3) and iy + 1 (taps rr = 2..5)
      float w0 = 0.f, w1 = 0.f;
      if (rr <= 3) w0 = (rr == 0 || rr == 3) ? 0.25f : ((rr == 1 && iy == 0) ? 1.f : 0.75f);
      if (rr >= 2) w1 = (rr == 2 || rr == 5) ? 0.25f : ((rr == 4 && iy + 2 == h) ? 1.f : 0.75f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0[i] += w0 * hx[i];
        acc1[i] += w1 * hx[i];
      }
    }
    float* o = gx + (p * h + iy) * w + ix0;
    if (add) {
      const float* q = add + (p * h + iy) * w + ix0;
      const float4 a0 = *reinterpret_cast<const float4*>(q), a1 = *reinterpret_cast<const float4*>(q + w);
      acc0[0] += a0.x, acc0[1] += a0.y, acc0[2] += a0.z, acc0[3] += a0.w;
      acc1[0] += a1.x, acc1[1] += a1.y, acc1[2] += a1.z, acc1[3] += a1.w;
    }
    *reinterpret_cast<float4*>(o) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
    *reinterpret_cast<float4*>(o + w) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Mask losses of the Hungarian-matched MaskFormer loss on the 2x up-sampled mask logits, without
// materialising them: for every matched prediction p (low-resolution logits pred[p] [h, w], binary target tgt[gt[p]] [2h, 2w])
//   u = bilinear2x(pred[p])  (F.interpolate, align_corners=False: mmdet/models/dense_heads/maskformer_head.py:475-479),
//   s = sigmoid(u),  sums[p] = { sum s*t, sum s, sum t, sum focal(u, t) }
// with focal(u, t) = BCEWithLogits(u, t) * (alpha t + (1 - alpha)(1 - t)) * ((1 - s) t + s (1 - t))^gamma
// (losses/focal_loss.py:36-44; the dice terms of losses/dice_loss.py:45-50 are formed from the first three sums).
// The reference runs ~15 element-wise passes over the [num_masks, 2h, 2w] tensor per decoder layer (16.7 ms forward at C2).
struct MaskPix {
  float s, t, bce, pt, at;
};

__device__ __forceinline__ MaskPix mask_pix(float u, unsigned char tb, float alpha) {
  MaskPix m;
  m.t = tb ? 1.f : 0.f;
  const float e = __expf(-fabsf(u));                     // exp(-|u|) in (0, 1]
  const float inv = 1.f / (1.f + e);
  m.s = u >= 0.f ? inv : e * inv;
  m.bce = fmaxf(u, 0.f) - u * m.t + log1pf(e);
  m.pt = (1.f - m.s) * m.t + m.s * (1.f - m.t);
  m.at = alpha * m.t + (1.f - alpha) * (1.f - m.t);
  return m;
}

// one thread: 4 consecutive hi-res pixels of one row per iteration; a workgroup walks `rows_per_wg` rows of one mask
__global__ __launch_bounds__(256) void mask_loss_fwd_kernel(const float* __restrict__ pred, const unsigned char* __restrict__ tgt,
                                                            const int64_t* __restrict__ gt, float* __restrict__ sums, int h,
                                                            int w, float alpha, float gamma, int chunks) {
  const int W = 2 * w, H = 2 * h;
  const int p = blockIdx.y;
  const float* pp = pred + (int64_t)p * h * w;
  const unsigned char* tp = tgt + gt[p] * (int64_t)H * W;
  const int quads = W / 4;
  const int64_t total = (int64_t)H * quads;
  const int64_t per = (total + chunks - 1) / chunks;
  const int64_t beg = blockIdx.x * per, end = beg + per < total ? beg + per : total;
  float a = 0.f, b = 0.f, c = 0.f, f = 0.f;
  for (int64_t idx = beg + threadIdx.x; idx < end; idx += 256) {
    const int q = (int)(idx % quads), oy = (int)(idx / quads);
    int y0, y1;
    float ly;
    taps(oy, h, y0, y1, ly);
    const float* r0 = pp + (int64_t)y0 * w;
    const float* r1 = pp + (int64_t)y1 * w;
    const uchar4 tv = *reinterpret_cast<const uchar4*>(tp + (int64_t)oy * W + q * 4);
    const unsigned char tb[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x0, x1;
      float lx;
      taps(q * 4 + j, w, x0, x1, lx);
      const float top = (1.f - lx) * r0[x0] + lx * r0[x1];
      const float bot = (1.f - lx) * r1[x0] + lx * r1[x1];
      const float u = (1.f - ly) * top + ly * bot;
      const MaskPix m = mask_pix(u, tb[j], alpha);
      a += m.s * m.t;
      b += m.s;
      c += m.t;
      f += m.bce * m.at * (gamma == 2.f ? m.pt * m.pt : __powf(m.pt, gamma));
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
    atomicAdd(sums + (int64_t)p * 4 + threadIdx.x,
              (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// gup[p][pixel] = d(sum_k g[p][k] * sums[p][k]) / du   (g[p][2], the weight of sum t, has no influence: t is data)
__global__ __launch_bounds__(256) void mask_loss_bwd_kernel(const float* __restrict__ pred, const unsigned char* __restrict__ tgt,
                                                            const int64_t* __restrict__ gt, const float* __restrict__ g,
                                                            float* __restrict__ gup, int h, int w, float alpha, float gamma,
                                                            int chunks) {
  const int W = 2 * w, H = 2 * h;
  const int p = blockIdx.y;
  const float* pp = pred + (int64_t)p * h * w;
  const unsigned char* tp = tgt + gt[p] * (int64_t)H * W;
  float* gp = gup + (int64_t)p * H * W;
  const float ga = g[p * 4], gb = g[p * 4 + 1], gf = g[p * 4 + 3];
  const int quads = W / 4;
  const int64_t total = (int64_t)H * quads;
  const int64_t per = (total + chunks - 1) / chunks;
  const int64_t beg = blockIdx.x * per, end = beg + per < total ? beg + per : total;
  for (int64_t idx = beg + threadIdx.x; idx < end; idx += 256) {
    const int q = (int)(idx % quads), oy = (int)(idx / quads);
    int y0, y1;
    float ly;
    taps(oy, h, y0, y1, ly);
    const float* r0 = pp + (int64_t)y0 * w;
    const float* r1 = pp + (int64_t)y1 * w;
    const uchar4 tv = *reinterpret_cast<const uchar4*>(tp + (int64_t)oy * W + q * 4);
    const unsigned char tb[4] = {tv.x, tv.y, tv.z, tv.w};
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x0, x1;
      float lx;
      taps(q * 4 + j, w, x0, x1, lx);
      const float top = (1.f - lx) * r0[x0] + lx * r0[x1];
      const float bot = (1.f - lx) * r1[x0] + lx * r1[x1];
      const float u = (1.f - ly) * top + ly * bot;
      const MaskPix m = mask_pix(u, tb[j], alpha);
      const float ds = m.s * (1.f - m.s);                                   // ds/du
      const float dpt = ds * (1.f - 2.f * m.t);                             // d pt / du
      const float ptg1 = gamma == 2.f ? m.pt : __powf(m.pt, gamma - 1.f);   // pt^(gamma-1)
      const float dfocal = m.at * ((m.s - m.t) * ptg1 * m.pt + m.bce * gamma * ptg1 * dpt);
      out[j] = (ga * m.t + gb) * ds + gf * dfocal;
    }
    *reinterpret_cast<float4*>(gp + (int64_t)oy * W + q * 4) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

inline int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int s2f_upsample2x_fwd(const float* x, float* y, int64_t planes, int h, int w, void* stream) {
  S2F_REQUIRE(x && y, S2F_EINVAL, "s2f_upsample2x_fwd: null pointer");
  S2F_REQUIRE(planes > 0 && h > 0 && w > 0 && (w % 2) == 0, S2F_EINVAL, "s2f_upsample2x_fwd: need even input width");
  S2F_REQUIRE(s2f_aligned16(y), S2F_EALIGN, "s2f_upsample2x_fwd: output must be 16-byte aligned");
  if ((w & 3) == 0 && s2f_aligned16(x))
    hipLaunchKernelGGL(up2x_fwd_block_kernel<false>, plane_grid((int64_t)h * (w / 4), planes), dim3(256), 0, (hipStream_t)stream, x, y,
                       planes, h, w);
  else
    hipLaunchKernelGGL(up2x_fwd_kernel, dim3(grid_for(planes * 2 * h * (2 * w / 4))), dim3(256), 0, (hipStream_t)stream, x, y,
                       planes, h, w);
  return s2f_check_launch("s2f_upsample2x_fwd");
}

extern "C" int s2f_upsample2x_sigmoid_fwd(const float* x, float* y, int64_t planes, int h, int w, void* stream) {
  S2F_REQUIRE(x && y, S2F_EINVAL, "s2f_upsample2x_sigmoid_fwd: null pointer");
  S2F_REQUIRE(planes > 0 && h > 0 && w > 0 && (w % 4) == 0, S2F_EINVAL, "s2f_upsample2x_sigmoid_fwd: needs w %% 4 == 0");
  S2F_REQUIRE(s2f_aligned16(x) && s2f_aligned16(y), S2F_EALIGN, "s2f_upsample2x_sigmoid_fwd: 16-byte alignment");
  hipLaunchKernelGGL(up2x_fwd_block_kernel<true>, plane_grid((int64_t)h * (w / 4), planes), dim3(256), 0, (hipStream_t)stream, x, y,
                     planes, h, w);
  return s2f_check_launch("s2f_upsample2x_sigmoid_fwd");
}

extern "C" int s2f_upsample2x_bwd_add(const float* gy, const float* add, float* gx, int64_t planes, int h, int w, void* stream) {
  S2F_REQUIRE(gy && gx, S2F_EINVAL, "s2f_upsample2x_bwd: null pointer");
  S2F_REQUIRE(planes > 0 && h > 0 && w > 0, S2F_EINVAL, "s2f_upsample2x_bwd: bad shape");
  if ((w & 3) == 0 && (h & 1) == 0 && s2f_aligned16(gy) && s2f_aligned16(gx) && s2f_aligned16(add))
    hipLaunchKernelGGL(up2x_bwd_block_kernel, plane_grid((int64_t)(h / 2) * (w / 4), planes), dim3(256), 0, (hipStream_t)stream,
                       gy, gx, planes, h, w, add);
  else
    hipLaunchKernelGGL(up2x_bwd_kernel, dim3(grid_for(planes * h * w)), dim3(256), 0, (hipStream_t)stream, gy, gx, planes, h, w, add);
  return s2f_check_launch("s2f_upsample2x_bwd");
}

extern "C" int s2f_upsample2x_bwd(const float* gy, float* gx, int64_t planes, int h, int w, void* stream) {
  return s2f_upsample2x_bwd_add(gy, nullptr, gx, planes, h, w, stream);
}

extern "C" int s2f_mask_loss_fwd(const float* pred, const uint8_t* tgt, const int64_t* gt_index, float* sums, int64_t P, int h,
                                 int w, float alpha, float gamma, void* stream) {
  if (P == 0) return S2F_OK;
  S2F_REQUIRE(pred && tgt && gt_index && sums, S2F_EINVAL, "s2f_mask_loss_fwd: null pointer");
  S2F_REQUIRE(P > 0 && P < 65536 && h > 0 && w > 0 && (w % 2) == 0, S2F_EINVAL, "s2f_mask_loss_fwd: need 0 < P < 65536, even w");
  S2F_REQUIRE((reinterpret_cast<uintptr_t>(tgt) & 3u) == 0, S2F_EALIGN, "s2f_mask_loss_fwd: targets must be 4-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (s2f_zero_async(sums, sizeof(float) * 4 * (size_t)P, s) != S2F_OK) return s2f_check_launch("s2f_mask_loss_fwd memset");
  const int64_t total = (int64_t)2 * h * (2 * w / 4);
  int chunks = (int)((total + 256 * 8 - 1) / (256 * 8));       // >= 8 iterations per thread: few atomics per mask
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(mask_loss_fwd_kernel, dim3(chunks, (unsigned)P), dim3(256), 0, s, pred, tgt, gt_index, sums, h, w, alpha,
                     gamma, chunks);
  return s2f_check_launch("s2f_mask_loss_fwd");
}

extern "C" int s2f_mask_loss_bwd(const float* pred, const uint8_t* tgt, const int64_t* gt_index, const float* g_sums, float* gup,
                                 int64_t P, int h, int w, float alpha, float gamma, void* stream) {
  if (P == 0) return S2F_OK;
  S2F_REQUIRE(pred && tgt && gt_index && g_sums && gup, S2F_EINVAL, "s2f_mask_loss_bwd: null pointer");
  S2F_REQUIRE(P > 0 && P < 65536 && h > 0 && w > 0 && (w % 2) == 0, S2F_EINVAL, "s2f_mask_loss_bwd: need 0 < P < 65536, even w");
  S2F_REQUIRE((reinterpret_cast<uintptr_t>(tgt) & 3u) == 0 && s2f_aligned16(gup), S2F_EALIGN,
              "s2f_mask_loss_bwd: targets 4-byte, gradient 16-byte aligned");
  const int64_t total = (int64_t)2 * h * (2 * w / 4);
  int chunks = (int)((total + 256 * 4 - 1) / (256 * 4));
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(mask_loss_bwd_kernel, dim3(chunks, (unsigned)P), dim3(256), 0, (hipStream_t)stream, pred, tgt, gt_index,
                     g_sums, gup, h, w, alpha, gamma, chunks);
  return s2f_check_launch("s2f_mask_loss_bwd");
}
