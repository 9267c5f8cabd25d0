/*
 * s2f.h -- C ABI of libs2f_hip.so: the MI355X (gfx950) kernels of the Spike2Former hot path.
 *
 * Conventions (every entry point):
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross this boundary.
 *   - all pointers are DEVICE pointers owned by the caller; nothing is allocated or freed inside;
 *     `?` in a comment marks a nullable pointer.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); launches are asynchronous.
 *   - return value: S2F_OK (0) or a negative S2F_E* code; `s2f_last_error()` gives the text.
 *   - re-entrant, no global state apart from the thread-local error string and the thread-local launch-timing arm.
 *
 * The reference (BICLab/Spike2Former) has no FFI on its live path -- everything is ATen ops called from
 * Python.  The in-tree precedents this ABI follows are the (dormant) pybind pair
 * `dcnv3_forward/backward` (ops_dcnv3/src/vision.cpp:14-17, argument list src/dcnv3.h:20-59) and the cupy
 * raw-pointer launch convention (Qtrick_architecture/clock_driven/cu_kernel_opt.py:53-71).  Each function
 * below names the reference code it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 */
#ifndef S2F_H
#define S2F_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2F_OK 0
#define S2F_EINVAL (-1)   /* bad argument (null pointer, non-positive size, unsupported geometry) */
#define S2F_EALIGN (-2)   /* pointer not aligned as required */
#define S2F_ELAUNCH (-3)  /* hipLaunch / runtime error */

#define S2F_ABI_VERSION 32
#define S2F_STAT_SLOTS 256

int s2f_version(void);
const char* s2f_last_error(void);

/* ---- measurement: kernel-timestamp timing of one call ------------------------------------------------
 * bench.py's roofline figures need the duration of individual launches.  Two hipEventRecord markers around a launch
 * add ~4 us of command-processor time to a ~10 us kernel; instead the launch itself carries the events
 * (hipExtLaunchKernelGGL start/stop events = the dispatch packet's own begin/end timestamps, what rocprofv3 reports).
 *   s2f_event_create/destroy: a hipEvent_t as void*.
 *   s2f_time_next_call(start, stop): arms THIS thread; the next s2f_lif_fwd / s2f_lif_bwd / s2f_bn_stats /
 *       s2f_bn_act_fwd / s2f_bn_act_bwd / s2f_spike_gemm_fwd / s2f_spike_gemm_dw / s2f_spike_conv3x3_* / s2f_split_gemm call stamps `start` with the begin of
 *       its first kernel and `stop` with the end of its last one, then disarms.  Not valid during stream capture.
 *   s2f_event_elapsed_us: stop - start in microseconds (both must have completed: synchronise first). */
void* s2f_event_create(void);
void s2f_event_destroy(void* event);
int s2f_time_next_call(void* start_event, void* stop_event);
int s2f_event_elapsed_us(void* start_event, void* stop_event, double* microseconds);

/* Number of uint64 words of the in-range bitmask for n elements: 4 words per 256-element tile.
 * Tile i covers elements [256 i, 256 i + 256); word j (0..3) bit l (0..63) belongs to element 256 i + 4 l + j
 * (lane l of a wavefront holds 4 consecutive elements; word j is the wave ballot of component j). */
int64_t s2f_lif_mask_words(int64_t n);

/* ---- a1/a2: Q_IFNode.forward + quant STE ------------------------------------------------------------
 * Replaces BaseNode.forward (Qtrick_architecture/clock_driven/neuron.py:166-197; charge :459-460, soft reset
 * :133-153) with `quant` (surrogate.py:522-538) as surrogate:
 *     h = v_in + x ; s = rint(clamp(h, 0, D)) (round-half-even) ; v_out = h - s*vth ; y = s / D
 *     mask bit = (0 <= h <= D)                          (what the STE backward needs instead of fp32 h)
 * v_in? NULL == membrane freshly reset (python float 0.).  v_out? NULL == membrane not kept.
 * mask? NULL == no backward wanted.  count_u8? NULL == integer counts not wanted.
 * stats? : uint64[S2F_STAT_SLOTS][2], atomically accumulated {sum of counts s, number of non-zero s}; the caller sums the
 *          slots (cal_firing_num.py:138-160 accumulates mean(y*D) = sum(stats[.][0])/n; firing_utils non-zero rate =
 *          sum(stats[.][1])/n).  Workgroup b adds into slot b % S2F_STAT_SLOTS: with ONE pair of counters the 2 048
 *          workgroups of a launch serialise on two L2 atomics (measured 52 us on a 5 us kernel).
 * x, y, v_* need 16-byte alignment when n >= 4.
 * y_bf16 != 0: y points to uint16 storage and the spikes are written as bf16 -- exact (a spike has <= 8 significant bits),
 *   2 bytes / element instead of the reference's 4; what the *_bf16 GEMM / attention entry points consume. */
int s2f_lif_fwd(const float* x, const float* v_in, void* y, float* v_out, uint64_t* mask, uint8_t* count_u8,
                uint64_t* stats, int64_t n, float vth, int D, int y_bf16, void* stream);

/* STE backward of one step:  gx = gv_out + (gy / D - gv_out * vth) * m   (gv_out? NULL == 0; dL/dv_in == gx). */
int s2f_lif_bwd(const float* gy, const float* gv_out, const uint64_t* mask, float* gx, int64_t n, float vth, int D,
                void* stream);
/* The same with the gradient sums of a fan-out folded in (each NULL = absent): gy2? = the gradient a SECOND consumer of the spike
 * map sends back, skip? = the gradient of a residual branch that read the neuron's INPUT (x + f(Q_IFNode(x)), sdtv2.py:203-219,
 * detr_layers.py:523-556):  gx = STE((gy + gy2) / D) + skip  -- the adds autograd's engine would launch on its own.  skip needs
 * gv_out == NULL. */
int s2f_lif_bwd_ports(const float* gy, const float* gy2, const float* gv_out, const uint64_t* mask, const float* skip, float* gx,
                      int64_t n, float vth, int D, void* stream);

/* Leaky charge in front of the same firing rule -- LIFNode.neuronal_charge (neuron.py:803-814) under the fork's BaseNode.forward
 * (:166-197: multi-level quantised firing, soft reset, y = s / D).  No Spike2Former module instantiates LIFNode; the
 * entry points exist so that the neuron file's second node type has the same kernel-backed mirror (host: neuron.LIFNode).
 *     decay_input != 0:  h = v + (x - v) / tau        (reference expression order, true fp32 division)
 *     decay_input == 0:  h = v * (1 - 1/tau) + x      (factor formed in double, rounded to fp32 once, as Python's scalar multiply)
 * then as s2f_lif_fwd.  v_in? NULL == membrane freshly reset (0.): h = x / tau resp. x.  tau > 1 (neuron.py:792).
 * Backward: g_h = gv_out + (gy / D - gv_out * vth) * m;  decay_input: gx = g_h / tau, gv_in = g_h - g_h / tau;
 * otherwise gx = g_h, gv_in = g_h * (1 - 1/tau).  gv_out? NULL == 0; gv_in? NULL == not wanted. */
int s2f_lif_leaky_fwd(const float* x, const float* v_in, void* y, float* v_out, uint64_t* mask, uint64_t* stats, int64_t n, float vth,
                      int D, float tau, int decay_input, int y_bf16, void* stream);
int s2f_lif_leaky_bwd(const float* gy, const float* gv_out, const uint64_t* mask, float* gx, float* gv_in, int64_t n, float vth, int D,
                      float tau, int decay_input, void* stream);

/* The decoder's value / key neurons on  a = x + e[c]  and  a + pos[b, c, l]  in one pass (x [TB, C, L] with tb = t*B + b,
 * e [C] = level_embed row, pos [B, C, L] = key positional encoding; mmdet/models/dense_heads/maskformer_head.py:535-540,
 * mmcv_spike/transformer.py:626-629): y_value = Q_IFNode(a), y_key = Q_IFNode(a + pos), reset stateless neurons, 1-bit
 * in-range masks as s2f_lif_fwd.  Backward: gx = STE(g_key, mask_key) + STE(g_value, mask_value); the gradient of e is the
 * per-channel sum of gx (left to the caller).  L % 4 == 0. */
int s2f_sum2_lif_fwd(const float* x, const float* e, const float* pos, void* y_key, void* y_value, uint64_t* mask_key,
                     uint64_t* mask_value, int64_t TB, int64_t B, int64_t C, int64_t L, float vth, int D, int y_bf16,
                     void* stream);
int s2f_sum2_lif_bwd(const float* g_key, const float* g_value, const uint64_t* mask_key, const uint64_t* mask_value,
                     float* gx, int64_t n, int D, void* stream);
/* The same with either incoming gradient optional (NULL = zero) and, in gx_key?, STE(g_key, mask_key) alone: summed over the T
 * time steps it is the gradient with respect to `pos` (the decoder's self- / cross-attention query neurons, whose position term is
 * the learnable query embedding: mmcv_spike/transformer.py:597-638). */
int s2f_sum2_lif_bwd_ex(const float* g_key, const float* g_value, const uint64_t* mask_key, const uint64_t* mask_value, float* gx,
                        float* gx_key, int64_t n, int D, void* stream);
/* ... and with a second consumer's gradient per spike map and a residual branch's gradient on x folded in (s2f_lif_bwd_ports):
 * the key / value spikes of one memory level are read by two decoder layers (maskformer_head.py:554-564), the decoder's
 * query + attention(query) reads the query beside its neurons (detr_layers.py:523-537). */
int s2f_sum2_lif_bwd_ports(const float* g_key, const float* g_key2, const float* g_value, const float* g_value2,
                           const uint64_t* mask_key, const uint64_t* mask_value, const float* skip, float* gx, float* gx_key, int64_t n,
                           int D, void* stream);

/* Layer scale folded into a BatchNorm's affine pair: w[c] = gamma[c] * s[c], b[c] = beta[c] * s[c]  (the pixel decoder's
 * `q + gamma_i * f(q)`, detr_layers.py:331-337, with f ending in a BatchNorm: u = s * BN(z) = BN_{gamma s, beta s}(z)).
 * Backward: dgamma = gw * s, dbeta = gb * s, ds = gw * gamma + gb * beta.  One launch each instead of 2 + 5 vector ops. */
int s2f_scale_affine_fwd(const float* gamma, const float* beta, const float* s, float* w, float* b, int C, void* stream);
int s2f_scale_affine_bwd(const float* gw, const float* gb, const float* gamma, const float* beta, const float* s,
                         float* dgamma, float* dbeta, float* ds, int C, void* stream);

/* ---- T successive stateful calls on one neuron, membrane carried in registers -----------------------
 * Same arithmetic as T calls of s2f_lif_fwd with v chained (what tools/cal_firing_num.py:203-225 does across
 * images; structural precedent: the dormant cupy kernel IFNode_fptt_softReset, neuron_kernel.py:19-119).
 * x_seq, y_seq: [T, n]; mask: [T, s2f_lif_mask_words(n)]; v0? / vT? : [n]. stats?: uint64[T][S2F_STAT_SLOTS][2]. */
int s2f_lif_seq_fwd(const float* x_seq, const float* v0, float* y_seq, float* vT, uint64_t* mask, uint64_t* stats,
                    int T, int64_t n, float vth, int D, void* stream);
/* BPTT: g_h[t] = g_h[t+1] + (gy[t]/D - g_h[t+1]*vth) * m[t], g_h[T] := gvT (NULL == 0); gx_seq[t] = g_h[t]; gv0? = g_h[0]. */
int s2f_lif_seq_bwd(const float* gy_seq, const float* gvT, const uint64_t* mask, float* gx_seq, float* gv0, int T,
                    int64_t n, float vth, int D, void* stream);

/* ---- BatchNorm fused with the conv bias, the residual add and the following Q_IFNode -------------------
 * Replaces the reference's chain  t = conv(x) + b ; u = BatchNorm(t) [+ residual] ; y = Q_IFNode(u)  (e.g. MS_ConvBlock,
 * SepConv, MS_MLP: mmseg/models/backbones/sdtv2.py:167-255; every Sequential(conv, BN) + Q_IFNode of the head), which
 * runs as ~12 ATen elementwise kernels forward and as many backward.  z: [N, C, L] channel-major, any L >= 1 (rows of
 * L % 4 != 0 elements take element-wise kernels: odd shapes only, e.g. a 10-query decoder; no map of the path at its real sizes).
 *
 * s2f_bn_stats (training only): per-channel sum / sum of squares of (z + conv_bias?) accumulated into sums_zeroed
 *   (double[2C], MUST be zero on entry -- the host hands out slices of one arena cleared once per step).
 * s2f_bn_act_fwd:  mean / rstd = 1/sqrt(var + eps) from `sums` (training) or from the running statistics (eval), written to
 *   stat_out[0:C] / stat_out[C:2C] (stat_out: float[3C]); training: running_mean?/running_var? updated in place with
 *   `momentum` and the unbiased variance, *num_batches_tracked? += 1 (torch.nn.BatchNorm semantics).
 *   stat_out[2C:3C] = beta - running_mean * gamma / sqrt(running_var + eps) with the UPDATED running statistics: the
 *   border value "BN(0)" of BNAndPadLayer (sdtv2.py:68-78), which s2f_dwconv_fwd takes as `border`.
 *   u = ((z + b) - mean) * rstd * gamma + beta [+ residual?] ; u_out? = u ;
 *   if y != NULL:  the Q_IFNode update of s2f_lif_fwd on u (v_in?, v_out?, mask?, stats? as there).
 * s2f_bn_act_bwd:  gu = g_u? + STE(g_y?, g_v?, mask) ;  training: gz = gamma*rstd*(gu - mean(gu) - xhat*mean(gu*xhat)),
 *   eval: gz = gamma*rstd*gu ;  g_residual? = gu ;  dgamma = sum(gu*xhat) ; dbeta = sum(gu).  sums_zeroed as above. */
/* 1 when, in training mode, s2f_bn_act_fwd / s2f_bn_act_bwd compute the channel statistics of this shape themselves
 * (one workgroup per channel holds the channel's N*L elements in registers: small maps with L % 256 == 0; one to sixteen wavefronts
 * per channel for rows that are not whole tiles, see s2f_bn_mask_words): the caller then skips s2f_bn_stats and may pass sums / sums_zeroed = NULL. */
int s2f_bn_single_pass(int64_t N, int64_t C, int64_t L);
/* uint64 words of the in-range mask that s2f_bn_act_fwd writes and s2f_bn_act_bwd reads for this shape in training mode: the flat
 * 256-element-tile layout of s2f_lif_mask_words, except for rows that are not whole tiles (L % 4 == 0, L % 256 != 0, C >= 32) with
 * N * L <= 20 480 -- the decoder's 100-token maps (N * L <= 2 048: one wavefront per channel, 32 words per channel) and C5's 50 x 84
 * maps (up to sixteen wavefronts per channel, 320 words per channel) -- whose single-pass kernels keep a per-channel mask layout of
 * their own (private to this forward / backward pair). */
int64_t s2f_bn_mask_words(int64_t N, int64_t C, int64_t L);
int s2f_bn_stats(const float* z, const float* conv_bias, double* sums_zeroed, int64_t N, int64_t C, int64_t L,
                 void* stream);
int s2f_bn_act_fwd(const float* z, const float* conv_bias, const double* sums, float* stat_out, float* running_mean,
                   float* running_var, int64_t* num_batches_tracked, const float* gamma, const float* beta,
                   const float* residual, float* u_out, const float* v_in, void* y, float* v_out, uint64_t* mask,
                   uint64_t* stats, int64_t N, int64_t C, int64_t L, float momentum, float eps, int training, float vth,
                   int D, int y_bf16, void* stream);
int s2f_bn_act_bwd(const float* z, const float* conv_bias, const float* stat, const float* gamma, const float* g_u,
                   const float* g_y, const float* g_v, const uint64_t* mask, double* sums_zeroed, float* gz,
                   float* g_residual, float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t L, int training,
                   float vth, int D, void* stream);

/* Train-mode BatchNorm o BatchNorm as ONE kernel (every RepConv chain of the attention blocks ends in two: Sequential(RepConv(..,
 * BN), BN), mmseg/models/backbones/sdtv2.py:112-132, 280-296, 304-306).  With xhat = (z + b - mean) r the first BatchNorm gives
 * a = gamma xhat + beta, whose batch mean is beta and whose biased batch variance is gamma^2 var r^2: the second needs no pass of
 * its own,  BN2(BN1(z)) = gamma gamma2 r2 xhat + beta2,  r2 = 1 / sqrt(gamma^2 var r^2 + eps2);  running_mean2 / running_var2 are
 * updated with (beta, gamma^2 var r^2 n/(n-1)).  Single-pass shapes only (s2f_bn2_fused_ok; the 32x32-stage maps where these
 * chains live); stat_out: float[4C] = mean, r, BN1(0) border, r2.  The backward returns both BatchNorms' parameter gradients:
 * dbeta2 = sum gu, dgamma2 = gamma r2 sum(gu xhat), dbeta = 0, dgamma = gamma2 eps2 r2^3 sum(gu xhat). */
int s2f_bn2_fused_ok(int64_t N, int64_t C, int64_t L);
int s2f_bn2_act_fwd(const float* z, const float* conv_bias, float* stat_out, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, const float* gamma, const float* beta, float momentum, float eps,
                    const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                    int64_t* num_batches_tracked2, float momentum2, float eps2, const float* residual, float* u_out,
                    const float* v_in, void* y, float* v_out, uint64_t* mask, uint64_t* stats, int64_t N, int64_t C, int64_t L,
                    float vth, int D, int y_bf16, void* stream);
int s2f_bn2_act_bwd(const float* z, const float* conv_bias, const float* stat, const float* gamma, const float* gamma2, float eps2,
                    const float* g_u, const float* g_y, const float* g_v, const uint64_t* mask, float* gz, float* g_residual,
                    float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, int64_t N, int64_t C, int64_t L, float vth, int D,
                    void* stream);

/* ---- im2col / col2im of the dense k x k convolutions that do not take the implicit 3x3 kernels: the stride-2 down-samplings and
 * the 7x7 stem (MS_DownSampling, mmseg/models/backbones/sdtv2.py:386-421).  cols [N][C kh kw][Ho Wo] (same dtype as x: fp32, or
 * bf16 spikes with x_bf16), row (c kh + ky) kw + kx, zero outside the plane, dilation 1 -- torch.nn.functional.unfold's layout.
 * s2f_col2im is its adjoint as a gather (no atomics, no zero fill): gx[n][c][y][x] = sum of the column entries that cover it. */
int s2f_im2col(const void* x, void* cols, int N, int C, int H, int W, int kh, int kw, int stride, int pad, int x_bf16, void* stream);
int s2f_col2im(const float* cols, float* gx, int N, int C, int H, int W, int kh, int kw, int stride, int pad, void* stream);

/* ---- depthwise KxK convolution (stride 1, dilation 1, K in {3,5,7}) on [N, C, H, W] -----------------------
 * Replaces nn.Conv2d(groups=C) as used by SepConv.dwconv (mmseg/models/backbones/sdtv2.py:156-163), RepConv's un-padded
 * 3x3 on the BNAndPadLayer output (sdtv2.py:48-89, 123-127), SepConv_Spike.dwconv (mmcv_spike/SNN_core.py:36-40),
 * DCNv3_pytorch.dw_conv (ops_dcnv3/modules/dcnv3.py:161-169) and the pixel decoder's output_convs (pixel_decoder.py:374-378).
 * w: [C, K, K].  Output size Ho = H + 2*pad - K + 1.  border?: per-channel value read inside the padding ring instead of
 * zero (BNAndPadLayer's constant border, sdtv2.py:68-84) -- the padded tensor is never materialised.
 * x_bf16 != 0: x is a bf16 spike map (uint16 storage) as the neuron kernels write it (SepConv.dwconv, the pixel decoder's
 * output convolutions and DCNv3's dw_conv all read a neuron output); the arithmetic stays fp32. */
int s2f_dwconv_fwd(const void* x, const float* w, const float* border, float* y, int N, int C, int H, int W, int K,
                   int pad, int x_bf16, void* stream);
/* Eval mode (row f4): the stencil with the BatchNorm (running statistics) and the Q_IFNode that follow every depthwise convolution of
 * the path in its store -- u_out? = fp32 pre-activation, y_bf16? = bf16 spikes, stats? = firing counters; the fp32 convolution
 * output is never written.  Per-element expressions of s2f_bn_act_fwd in eval mode (bit-identical spikes). */
int s2f_dwconv_bn_lif_fwd(const void* x, const float* w, const float* border, const float* running_mean, const float* running_var,
                          const float* gamma, const float* beta, float eps, float* u_out, void* y_bf16, uint64_t* stats, int N,
                          int C, int H, int W, int K, int pad, int x_bf16, float vth, int D, void* stream);
int s2f_dwconv_bwd_input(const float* gy, const float* w, float* gx, int N, int C, int H, int W, int K, int pad,
                         void* stream);
int s2f_dwconv_bwd_weight(const void* x, const float* border, const float* gy, float* gw, int N, int C, int H, int W,
                          int K, int pad, int accumulate, int x_bf16, void* stream);

/* ---- spike GEMM on the bf16 matrix cores ---------------------------------------------------------------
 * Y[b] (M x N) = W (M x K) @ X[b] (K x N) [+ bias[M]]   for b in [0, batch): the 1x1 Conv2d / Conv1d(k=1) / im2col'd kxk
 * convolution of the path on channel-major activations [batch, K, N] whose input is a Q_IFNode output (q/k/v/proj RepConv
 * 1x1, sdtv2.py:121-125, 304-306; MS_MLP / MS_ConvBlock / MS_DownSampling, sdtv2.py:197-204, 229-235, 399-405; the head's
 * Conv1d / 1x1 Conv2d, mmcv_spike/transformer.py:213-236, 758-763, pixel_decoder.py:368-404, SNN_core.py:31-45).
 * X must hold spikes -- multiples of 1/D with at most 8 significant bits (exact in bf16).  W is pre-split by
 * s2f_split_bf16x3 into hi + mid + lo bf16 terms (24 mantissa bits); products are exact, accumulation is fp32:
 * the accuracy of an fp32 GEMM on v_mfma_f32_32x32x16_bf16.  `terms` in {1,2,3} = number of weight terms used.
 * w_split: [3][Mpad][Kpad] bf16, zero padded, Mpad % 64 == 0, Kpad % 32 == 0.  N % 4 == 0. */
int s2f_split_bf16x3(const float* w, uint16_t* w_split, int M, int K, int Mpad, int Kpad, void* stream);
/* All weights of a model in ONE launch -- the re-split a training step owes after the optimiser has updated them, and the
 * node a captured step replays so that every replay multiplies by the LIVE fp32 weights.  jobs (device): int64 [njobs][8] =
 * {src fp32 pointer, dst bf16 pointer, M, K, Mpad, Kpad, mode | (C << 8), first workgroup}; job i owns workgroups
 * [first_i, first_i + ceil(Mpad Kpad / 1024)), total_workgroups = their sum.  mode 0: src = the [M][K] matrix; 1: src = a
 * conv weight [M][C][3][3] read tap-major (the w_split layout of s2f_spike_conv3x3_fwd); 2: the flipped, transposed
 * matrix of a conv weight [C][M][3][3] that s2f_conv3x3_general takes for the input gradient (C field = its out-channels). */
int s2f_split_bf16x3_multi(const int64_t* jobs, int njobs, int64_t total_workgroups, void* stream);
int s2f_spike_gemm_fwd(const uint16_t* w_split, const float* X, const float* bias, float* Y, int batch, int M, int N,
                       int K, int Mpad, int Kpad, int terms, void* stream);
/* 3x3 convolution (stride 1, padding 1) of a SPIKE activation as an implicit GEMM on the same kernels: the B operand is
 * the activation X [batch, C, H, W] itself, the kernels' loaders form the rows of the im2col matrix on the fly -- the 9x
 * inflated column matrix of F.unfold is never materialised (MS_ConvBlock.conv1 / conv2,
 * mmseg/models/backbones/sdtv2.py:197-204).  The contraction runs TAP-MAJOR, k = (3 ky + kx) C + c (C % 32 == 0: a 32-wide
 * step lies in one tap): w_split = s2f_split_bf16x3 of weight.permute(0,2,3,1) viewed [M, 9C].  Y: [batch, M, H*W].
 * s2f_spike_conv3x3_fwd needs W % 4 == 0; s2f_spike_conv3x3_dw needs W % 4 == 0 (round 5: any such width; a power of two decodes the pixel index with a shift) and writes dW in the same tap-major
 * order, [M, 3, 3, C] (dY [batch, M, H*W]); the caller permutes it back to the weight's [M, C, 3, 3]. */
int s2f_spike_conv3x3_fwd(const uint16_t* w_split, const float* X, const float* bias, float* Y, int batch, int M, int C,
                          int H, int W, int Mpad, int Kpad, int terms, void* stream);
int s2f_spike_conv3x3_dw(const float* dY, const float* X, float* dW, int batch, int M, int C, int H, int W, int accumulate,
                         void* stream);
/* 3x3 convolution (stride 1, padding 1) of a GENERAL fp32 tensor, implicit GEMM with both operands split hi+mid+lo
 * (6 passes): Y[b] (M x H*W) = Wt (M x 9C, tap-major as above, Mpad % 128 == 0) (*) X[b] ([C, H, W]).  Used for the input
 * gradient of the 3x3 convolutions as the transposed convolution dX = flip(W)^T (*) dY -- no unfold(dY), no col2im. */
int s2f_conv3x3_general(const uint16_t* w_split, const float* X, float* Y, int batch, int M, int C, int H, int W, int Mpad,
                        int Kpad, void* stream);
/* General split GEMM on the same kernel structure:  Y[b] (M x N) = out_scale * A[b] (M x K) @ X[b] (K x N).
 *   a_split: bf16 terms as written by s2f_split_bf16x3: term i of batch b starts at a_split + i*a_term_stride +
 *     b*a_batch_stride (ELEMENTS; rows Kpad apart, Mpad readable rows per batch; a_batch_stride = 0 shares one A; several
 *     batches may be row blocks of ONE split matrix); a_terms in {1, 3} = terms used (1: A exact in bf16, e.g. spikes).
 *   X: fp32; row k of batch b lies at X + b*x_batch_stride + (k / k_inner)*x_outer_stride + (k % k_inner)*N (floats): a
 *     contraction over the T slabs of a [T, B, C, HW] tensor is ONE launch with K = T*C, k_inner = C, x_outer_stride =
 *     B*C*HW.  x_terms in {1, 3}: X is split in the kernel into that many bf16 terms.
 *   (a_terms, x_terms) = (1,3) or (3,1): 3 MFMA passes, exact products; (3,3): 6 passes (terms i + j <= 2), 2^-24.
 *   Mpad % 128 == 0, Kpad % 32 == 0, N % 4 == 0.  Call sites: the mask einsum 'ltbqc,tbchw->ltbqhw' + mean over t
 *   (mmdet/models/dense_heads/maskformer_head.py:582-583) forward and its d(mask_features). */
int s2f_split_gemm(const uint16_t* a_split, int64_t a_batch_stride, int64_t a_term_stride, int a_terms, const float* X,
                   int64_t x_batch_stride,
                   int k_inner, int64_t x_outer_stride, int x_terms, float* Y, int64_t y_batch_stride, float out_scale,
                   int batch, int M, int N, int K, int Mpad, int Kpad, void* stream);
/* Weight gradient of the same convolution:  dW[m][k] = sum_b sum_l dY[b][m][l] * X[b][k][l]   (dY [batch, M, L],
 * X [batch, K, L] spikes, dW [M, K] overwritten).  X is exact in bf16; dY is split on the fly into hi + mid + lo bf16
 * terms: three MFMA passes, exact products, fp32 accumulation (split-K partial tiles are combined with fp32 atomics, so
 * the result is reproducible to fp32 round-off, not bit for bit).  L % 4 == 0.
 * accumulate != 0: dW += ... (no clearing memset; for gradients summed straight into a pre-zeroed flat buffer).
 * x_terms: 1 = X exact in bf16 (spikes), 3 passes; 3 = X a general fp32 tensor, split like dY: 6 passes (terms i + j <= 2)
 *   -- the "both operands contraction-contiguous" GEMM of two general tensors, e.g. dE = g @ MF^T of the mask einsum. */
int s2f_spike_gemm_dw(const float* dY, const float* X, float* dW, int batch, int M, int K, int L, int accumulate, int x_terms,
                      void* stream);

/* ---- the same GEMMs with the activation operand ARRIVING in bf16 ------------------------------------------------------
 * The neuron kernels can write their spikes as bf16 (`y_bf16` of s2f_lif_fwd / s2f_bn_act_fwd / s2f_sum2_lif_fwd): spikes are
 * multiples of 1/D with at most 8 significant bits, so bf16 holds them exactly, at half the bytes of the reference's fp32
 * tensors.  X: bf16 [batch, K, N] (uint16_t storage), channel-major as above.  The K loops then hold loads, LDS traffic and
 * MFMAs only: the forward kernel copies the X tile as it lies in memory and forms the k-contiguous MFMA fragments with the
 * LDS transpose read (ds_read_b64_tr_b16); the weight-gradient kernel reads the contraction-contiguous rows directly.
 * Same arithmetic, same results as the fp32-operand entry points above.  N % 4 == 0 (16-byte chunks when N % 8 == 0).
 * s2f_to_bf16_exact: fp32 -> bf16 for a tensor that is exactly representable (spikes a caller still holds in fp32); n % 4 == 0. */
int s2f_to_bf16_exact(const float* x, uint16_t* y, int64_t n, void* stream);
int s2f_spike_gemm_fwd_bf16(const uint16_t* w_split, const uint16_t* X, const float* bias, float* Y, int batch, int M, int N,
                            int K, int Mpad, int Kpad, int terms, void* stream);
int s2f_spike_conv3x3_fwd_bf16(const uint16_t* w_split, const uint16_t* X, const float* bias, float* Y, int batch, int M,
                               int C, int H, int W, int Mpad, int Kpad, int terms, void* stream);
/* General form of the forward kernel: Y[b] (M x N) = out_scale * (A[b] (M x K) @ X[b] (K x N) + bias[b][m]) with a weight per
 * batch element (a_split + b * a_batch_stride: three bf16 terms [3][Mpad][Kpad] per batch, Mpad * Kpad * 3 apart at least),
 * and the contraction running over K / k_inner slabs of a bf16 spike map that lie x_outer_stride elements apart (row k of
 * batch b at X + b * x_batch_stride + (k / k_inner) * x_outer_stride + (k % k_inner) * N; k_inner % 32 == 0).  Call site:
 * the mask contraction of the head with the mask_feature 1x1 convolution folded into it (ops.mask_einsum_folded):
 * sum_t (E_t W_mf) @ S_t over the T time slices of mask_feature_spike's output as ONE contraction of length T*C
 * (mmdet/models/dense_heads/maskformer_head.py:582-583 + mmdet/models/layers/pixel_decoder.py:467-470). */
int s2f_spike_gemm_fwd_bf16_ex(const uint16_t* a_split, int64_t a_batch_stride, const uint16_t* X, int64_t x_batch_stride,
                               int k_inner, int64_t x_outer_stride, const float* bias, int64_t bias_batch_stride,
                               float out_scale, float* Y, int batch, int M, int N, int K, int Mpad, int Kpad, void* stream);
/* The same general product on the LDS-DMA pipeline (csrc/pgemm.hip, round 5): a_pack holds one s2f_pack_bf16x3 pack of [M][K] per
 * batch element, a_batch_stride elements apart; N % 8 == 0. */
int s2f_pgemm_nn_bf16_ex(const uint16_t* a_pack, int64_t a_batch_stride, const uint16_t* X, int64_t x_batch_stride, int k_inner,
                         int64_t x_outer_stride, const float* bias, int64_t bias_batch_stride, float out_scale, float* Y, int batch,
                         int M, int N, int K, void* stream);
int s2f_spike_gemm_dw_bf16(const float* dY, const uint16_t* X, float* dW, int batch, int M, int K, int L, int accumulate,
                           void* stream);
/* Implicit 3x3 weight gradient (stride 1, padding 1).  accumulate: bit 0 = add into dW (otherwise dW is zeroed first); bit 1 = dW in
 * the WEIGHT's layout [M][C][3][3] instead of the tap-major [M][3][3][C] the kernel contracts in -- with both bits the gradient goes
 * straight into the parameter's slot of the flat gradient buffer (no zeroed staging tensor, no permuted add). */
int s2f_spike_conv3x3_dw_bf16(const float* dY, const uint16_t* X, float* dW, int batch, int M, int C, int H, int W,
                              int accumulate, void* stream);
/* MANY weight gradients in one launch.  jobs: HOST array of njobs (<= 56) records {dY, X, dW (device pointers), batch, M, K,
 * L} as int64; each is the product of s2f_spike_gemm_dw_bf16 with accumulate != 0 (dW += ...: the destinations are slices
 * of a pre-zeroed flat gradient buffer).  The short-contraction layers of the path (32x32 / 64x64 stages, the decoder's
 * 100-token layers: ~180 launches per step of 18-35 us each, every one split 32-64 ways to fill the chip) become one grid
 * over all (job, tile, split) triples with a launch-wide contraction length per workgroup; the job table travels in the
 * kernel arguments.  bkv = contraction elements per step: 64 (rows with L % 64 == 0 or L >= 512) or 32. */
int s2f_spike_gemm_dw_grouped(const int64_t* jobs, int njobs, int bkv, void* stream);
/* The same weight gradients on the LDS-DMA pipeline (csrc/dwp.hip, round 5): tile 128 (dY rows) x 256 (X rows), contraction step
 * 32, eight wavefronts in two halves that alternate a multiply segment with a staging segment (X by global_load_lds into a
 * three-slot ring, dY split hi + mid + lo through registers into two stages), one barrier per segment.  Needs L % 4 == 0,
 * L >= 32, M L < 2^30, K L < 2^31; with L % 32 != 0 the last step of a batch element is masked while dY is split and the copies
 * are range-checked against the whole tensors: batch M L < 2^30, batch K L < 2^31 (s2f_spike_gemm_dw_pipe_ok says whether a shape qualifies; the host falls back to
 * s2f_spike_gemm_dw_bf16 otherwise).  cfg: 0 = the two-halves schedule, 1 = every wavefront in the same phase (probe);
 * target_wgs <= 0: default split of the contraction.  Replaces the autograd weight gradient of the 1x1 convolutions fed by a
 * Q_IFNode (mmseg/models/backbones/sdtv2.py:222-255, 304-306; mmcv_spike/transformer.py:213-236, 758-763;
 * mmdet/models/layers/pixel_decoder.py:368-404; mmdet/models/dense_heads/maskformer_head.py:581-582 for the mask contraction's
 * embedding gradient). */
int s2f_spike_gemm_dw_pipe_ok(int batch, int M, int K, int L);
int s2f_spike_gemm_dw_pipe(const float* dY, const uint16_t* X, float* dW, int batch, int M, int K, int L, int accumulate, int cfg,
                           int target_wgs, void* stream);
int s2f_spike_gemm_dw_pipe_grouped(const int64_t* jobs, int njobs, int cfg, int target_wgs, void* stream);
/* The implicit 3x3 weight gradient (stride 1, padding 1; what s2f_spike_conv3x3_dw_bf16 computes: dW [M][3][3][C] tap-major) on the same
 * kernel, up to 16 convolutions per launch.  The shifted rows of a tap are 2-byte aligned, which an LDS-DMA cannot fetch: the kernel
 * reads the horizontal taps from Xs, a copy of the activation shifted by one element behind an 8-element front pad (s2f_shift1_bf16:
 * Xs holds n + 16 elements, Xs[8 + i] = X[i + 1] for i = -1 .. n - 2, zeros elsewhere; n % 8 == 0),
 * and zeroes in the fragments what the zero padding would have supplied.  jobs (HOST array): njobs x {dY, X, Xs, dW (pointers),
 * batch, M, C, H, W} as int64; every dW is accumulated into; cfg bit 0 = the probe schedule of s2f_spike_gemm_dw_pipe, bit 1 = dW in
 * the weight's layout [M][C][3][3] (as s2f_spike_conv3x3_dw_bf16's accumulate bit 1).  Needs C % 32 == 0, W % 8 == 0, W >= 32, H W % 32 == 0, B C H W 2 < 2^31
 * (s2f_spike_conv3x3_dw_pipe_ok).  Replaces the autograd weight gradient of MS_ConvBlock's dense 3x3 convolutions and of the stride-1
 * down-sampling (mmseg/models/backbones/sdtv2.py:183-219, 540-548). */
int s2f_spike_conv3x3_dw_pipe_ok(int batch, int M, int C, int H, int W);
int s2f_shift1_bf16(const uint16_t* X, uint16_t* Xs, int64_t n, void* stream);
int s2f_spike_conv3x3_dw_pipe(const int64_t* jobs, int njobs, int cfg, int target_wgs, void* stream);

/* ---- pipelined GEMMs fed by LDS-DMA (csrc/pgemm.hip, round 3) ------------------------------------------------------------
 * The same products as s2f_spike_gemm_fwd_bf16 and as the autograd input gradient of a 1x1 convolution
 * (mmseg/models/backbones/sdtv2.py:121-125, 164, 222-255, 304-306; mmcv_spike/transformer.py:196-361), with the operand tiles
 * copied global -> LDS asynchronously (global_load_lds_dwordx4), 2-3 LDS stages and one barrier per K step.
 * The weight operand is PRE-PACKED into the kernels' LDS image: blocks of [3 terms hi|mid|lo][64 rows][32 k] bf16, block
 * (mb, kb) at element ((mb * ceil(K/32) + kb) * 6144), the 16-byte chunk c of row r stored at chunk c ^ ((r >> 2) & 3), rows
 * and columns past M / K zero.  s2f_pack_elems(M, K) = bf16 elements of a pack.
 * s2f_pack_bf16x3_multi: every pack of a model in one launch (what a training step owes after the optimiser update; recorded
 *   in a captured step).  jobs (device): int64 [njobs][8] = {src fp32 pointer, dst bf16 pointer, M, K, mode | (C << 8), first
 *   workgroup, 0, 0}; job i owns workgroups [first_i, first_i + ceil(ceil(M/64) ceil(K/32) 2048 / 1024)).  mode 0: src is the
 *   [M][K] matrix; 1: a conv weight [M][C][3][3] read tap-major (k = tap C + c); 2: the flipped, transposed matrix of a conv
 *   weight [Mw][M][3][3] (C field = Mw): A[c][tap Mw + m] = src[(m M + c) 9 + 8 - tap]; 3: the transpose of a [K][M] matrix.
 * s2f_pack_bf16x3: one matrix (the job in the kernel arguments: usable inside a stream capture).
 * s2f_pgemm_nn_bf16:  Y[b] (M x N) = A (M x K, packed) @ X[b] (K x N bf16 spikes, n contiguous) [+ bias[m]];  N % 8 == 0;
 *   terms in {1,2,3}; cfg: 0 = pick a tile by the problem size, 1..5 = force one (probe switch).
 * s2f_pgemm_dx_f32:   DX[b] (Ki x N) = W^T (Ki x Mo) @ G[b] (Mo x N, fp32) [+ beta * DX[b]] with W (Mo x Ki) given as ITS
 *   forward pack (the same buffer the forward product reads); G is split hi + mid + lo in the kernel: 6 MFMA passes, 2^-24.
 *   N % 4 == 0; batch strides in elements (0 = dense).  With the pack of W^T (mode 3) the same kernel is the forward product
 *   Y = W @ X of a convolution whose input X is a general fp32 tensor.  Replaces the library fp32 GEMM (rocBLAS / hipBLASLt) of round 1-2. */
/* Eval-mode fusion (the reference's fold: Qtrick_architecture/clock_driven/functional.py:574-692):
 *   y = Q_IFNode( BatchNorm_running( W @ X + conv_bias ) [+ residual] )   as ONE launch -- the GEMM above with bn_apply's and
 * the neuron's arithmetic in its epilogue (same per-element expressions: bit-identical to s2f_pgemm_nn_bf16 + s2f_bn_act_fwd in
 * eval mode).  u_out? = the fp32 pre-activation, y_bf16? = the spikes (bf16), v_in? / v_out? = carried membrane, stats? = firing
 * counters as in s2f_lif_fwd.  N % 4 == 0, N >= 8; the tile follows the plain product's rule (round 4: the eight-wavefront
 * register-staged tiles too).  No backward (inference path). */
int s2f_gemm_bn_lif_fwd(const uint16_t* a_pack, const uint16_t* X, const float* conv_bias, const float* running_mean,
                        const float* running_var, const float* gamma, const float* beta, float eps, const float* residual,
                        float* u_out, const float* v_in, void* y_bf16, float* v_out, uint64_t* stats, int batch, int M, int N,
                        int K, float vth, int D, void* stream);
/* The same epilogue on the implicit 3x3 convolution (stride 1, padding 1; w_pack = the tap-major pack, s2f_pack_bf16x3 mode 1; X
 * [batch][C][H][W] bf16 spikes): eval-mode conv3x3 -> BatchNorm [+ residual] [-> Q_IFNode] as one launch (MS_ConvBlock's two
 * convolutions in inference, mmseg/models/backbones/sdtv2.py:183-219). */
int s2f_conv3x3_bn_lif_fwd(const uint16_t* w_pack, const uint16_t* X, const float* conv_bias, const float* running_mean,
                           const float* running_var, const float* gamma, const float* beta, float eps, const float* residual,
                           float* u_out, const float* v_in, void* y_bf16, float* v_out, uint64_t* stats, int batch, int M, int C,
                           int H, int W, float vth, int D, void* stream);
int64_t s2f_pack_elems(int M, int K);
int s2f_pack_bf16x3(const float* src, uint16_t* dst, int M, int K, int mode, int C, void* stream);
int s2f_pack_bf16x3_multi(const int64_t* jobs, int njobs, int64_t total_workgroups, void* stream);
int s2f_pgemm_nn_bf16(const uint16_t* a_pack, const uint16_t* X, const float* bias, float* Y, int batch, int M, int N, int K,
                      int terms, int cfg, void* stream);
int s2f_pgemm_dx_f32(const uint16_t* w_pack, const float* G, int64_t g_batch_stride, float* DX, int64_t dx_batch_stride,
                     int batch, int Mo, int Ki, int N, float beta, int cfg, void* stream);
/* Inference: 1x1 convolution of a DENSE fp32 input -> BatchNorm (running statistics) [+ residual] [-> Q_IFNode from a reset
 * membrane] as ONE launch, `groups` (1..4) independent weights on consecutive channel groups: SepConv.pwconv2 behind the depthwise
 * convolution (sdtv2.py:135-180), the second 1x1 of the RepConv chains with their composed BatchNorm pair (sdtv2.py:112-132,
 * 304-306; the fold helpers of clock_driven/functional.py:574-692), the stem's column matrix (sdtv2.py:386-421).
 * X [batch][groups K][N] fp32 (strides in elements), per-channel vectors [groups M], residual / u_out / y_bf16
 * [batch][groups M][N]; w_packs[g] = s2f_pack_bf16x3 (mode 3) of W_g [M][K].  Six bf16 passes (fp32 accuracy).  N % 4 == 0. */
int s2f_dense_gemm_bn_lif_fwd(const uint16_t* const* w_packs, int groups, const float* X, int64_t x_batch_stride, int64_t x_group_stride,
                              const float* conv_bias, const float* running_mean, const float* running_var, const float* gamma,
                              const float* beta, float eps, const float* residual, float* u_out, void* y_bf16, uint64_t* stats,
                              int batch, int K, int M, int N, float vth, int D, void* stream);
/* Up to four independent products of s2f_pgemm_dx_f32's plain-store form in ONE launch (blockIdx.z = group): w_packs is a HOST
 * array of `groups` device pointers (copied into the kernel arguments); group g reads G + g * g_group_stride and writes DX + g *
 * dx_group_stride (and bn_partials? + g * partials_group_stride: the _stats form).  The channel groups of a grouped 1x1
 * convolution -- the second 1x1 of the stacked q | k | v projection chain (sdtv2.py:304-306), forward and input gradient. */
int s2f_pgemm_dx_f32_grouped(const uint16_t* const* w_packs, int groups, const float* G, int64_t g_batch_stride,
                             int64_t g_group_stride, float* DX, int64_t dx_batch_stride, int64_t dx_group_stride, float* bn_partials,
                             int64_t partials_group_stride, int batch, int Mo, int Ki, int N, void* stream);
/* Implicit 3x3 convolution (stride 1, padding 1) on the same pipeline: Y[b] (M x H W) = A (M x 9 C, the TAP-MAJOR pack of the
 * weight: s2f_pack_bf16x3 mode 1) @ im2col(X[b]), the column matrix never formed.  _bf16: X [batch][C][H][W] bf16 spikes, 3 passes
 * (forward; replaces s2f_spike_conv3x3_fwd_bf16, bit-identical).  _f32: X fp32, split in the kernel, 6 passes -- the INPUT
 * gradient as the convolution of dY [batch][C = conv out_channels][H][W] with the flipped, transposed weight (pack mode 2, M =
 * conv in_channels; replaces s2f_conv3x3_general, bit-identical).  C % 32 == 0, W % 4 == 0.  cfg 0 = automatic tile. */
int s2f_pgemm_conv3x3_bf16(const uint16_t* w_pack, const uint16_t* X, const float* bias, float* Y, int batch, int M, int C, int H,
                           int W, int cfg, void* stream);
int s2f_pgemm_conv3x3_f32(const uint16_t* w_pack, const float* X, float* Y, int batch, int M, int C, int H, int W, int cfg,
                          void* stream);
/* BatchNorm statistics from the producing GEMM's epilogue, without atomics (round 4; reference chain conv -> BatchNorm ->
 * Q_IFNode: mmseg/models/backbones/sdtv2.py:222-255, 304-333).  The `_stats` forms of the three forward
 * products above (no bias, all three weight terms, automatic tile) additionally store, per output row and per workgroup tile
 * (128 columns of one batch element), the fp32 sum and sum of squares of the tile's values:
 *   bn_partials[(row * P + p) * 2 + {0, 1}],   p = b * ceil(N / 128) + column tile,   P = s2f_bn_partials_count(batch, N)
 * (channel-major: one channel's partials are contiguous; plain stores: no atomics, nothing to zero, deterministic).  A group of a
 * grouped product (s2f_pgemm_dx_f32_stats with batch strides) writes its rows into a wider table: pass bn_partials + 2 * first_row * P.
 * s2f_bn_partials_finalize adds the P partials of every channel in fp64 in a fixed order into sums_out (double[2C], need not be
 * zeroed) -- the sums s2f_bn_stats produces (of z + conv_bias?: the shift is applied to the sums), which s2f_bn_act_fwd then takes
 * as `sums`.  The statistics pass over z -- one read of the tensor per BatchNorm, 5-47 us at C2 -- becomes a 2.5-4 us launch over
 * P * C * 8 bytes, and the statistics no longer depend on the order of fp64 atomics. */
int64_t s2f_bn_partials_count(int batch, int N);
int s2f_pgemm_nn_bf16_stats(const uint16_t* a_pack, const uint16_t* X, float* Y, float* bn_partials, int batch, int M, int N, int K,
                            void* stream);
int s2f_pgemm_conv3x3_bf16_stats(const uint16_t* w_pack, const uint16_t* X, float* Y, float* bn_partials, int batch, int M, int C,
                                 int H, int W, void* stream);
int s2f_pgemm_dx_f32_stats(const uint16_t* w_pack, const float* G, int64_t g_batch_stride, float* DX, int64_t dx_batch_stride,
                           float* bn_partials, int batch, int Mo, int Ki, int N, void* stream);
int s2f_bn_partials_finalize(const float* partials, int64_t P, const float* conv_bias, double* sums_out, int64_t N, int64_t C,
                             int64_t L, void* stream);
/* s2f_bn_act_bwd with a SECOND gradient of the spike map (g_y2?, NULL = absent): the neuron fused into this BatchNorm has two
 * readers (the backbone taps feed the next stage and the pixel decoder's lateral convolution, mask2former's pixel_decoder.py:316-472;
 * DCNv3's offset and mask convolutions share offset_spike, dcnv3.py:209-214), and g_y + g_y2 is formed where g_y is read instead of
 * by an add launch of the autograd engine.  Only on the kernels s2f_bn_bwd_ports_ok() names (1 = supported for this shape). */
int s2f_bn_bwd_ports_ok(int64_t N, int64_t C, int64_t L, int training, int D);
int s2f_bn_act_bwd_ports(const float* z, const float* conv_bias, const float* stat, const float* gamma, const float* g_u,
                         const float* g_y, const float* g_y2, const float* g_v, const uint64_t* mask, double* sums_zeroed, float* gz,
                         float* g_residual, float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t L, int training, float vth,
                         int D, void* stream);

/* The gradient of a GEMM-produced pre-activation as THREE bf16 PLANES hi | mid | lo (gz = hi + mid + lo to 2^-24; plane p at
 * gz_split + p * N C L): s2f_bn_act_bwd_split is s2f_bn_act_bwd writing that form instead of fp32 (6 instead of 4 bytes per
 * element), s2f_pgemm_dx_split the input-gradient product reading it (plane p of batch b at G_split + p * plane_stride +
 * b * Mo * N): both operands of its K loop arrive by LDS-DMA, nothing is converted or stored to LDS by the wavefronts.
 * N % 8 == 0. */
int s2f_bn_act_bwd_split(const float* z, const float* conv_bias, const float* stat, const float* gamma, const float* g_u,
                         const float* g_y, const float* g_v, const uint64_t* mask, double* sums_zeroed, uint16_t* gz_split,
                         float* g_residual, float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t L, int training,
                         float vth, int D, void* stream);
int s2f_pgemm_dx_split(const uint16_t* w_pack, const uint16_t* G_split, int64_t plane_stride, float* DX, int batch, int Mo,
                       int Ki, int N, int cfg, void* stream);
/* The weight gradients of the same layers from the planes: s2f_spike_gemm_dw_bf16 / s2f_spike_gemm_dw_grouped with dY given as
 * hi | mid | lo (plane_stride >= batch * M * L elements apart; in the grouped form every job's dY is such a triple, batch * M * L
 * apart): the tile is copied into LDS as it lies -- the loop converts nothing. */
int s2f_spike_gemm_dw_bf16_split(const uint16_t* dY_split, int64_t plane_stride, const uint16_t* X, float* dW, int batch, int M,
                                 int K, int L, int accumulate, void* stream);
int s2f_spike_gemm_dw_grouped_split(const int64_t* jobs, int njobs, int bkv, void* stream);
/* dW (+)= sum_b dY[b] (M x L) X[b]^T (L x K), both operands general fp32 (6 passes), batch strides in elements (0 = dense);
 * accumulate: bit 0 = add into dW (else it is zeroed first), bit 1 = no contraction split (one add per element: run-to-run identical):
 * the weight gradient of the 1x1 convolutions whose input is not a spike map. */
int s2f_gemm_dw_general(const float* dY, int64_t dy_batch_stride, const float* X, int64_t x_batch_stride, float* dW, int batch,
                        int M, int K, int L, int accumulate, void* stream);
/* Many of them in ONE launch (the counterpart of s2f_spike_gemm_dw_grouped for two general operands): jobs = HOST array of
 * njobs x {dY, dy_batch_stride, X, x_batch_stride, dW, batch, M, K, L} (int64 each; strides in elements, 0 = dense); every dW
 * is ACCUMULATED into.  1 <= njobs <= 56; the table travels in the kernel arguments (capturable). */
int s2f_gemm_dw_general_grouped(const int64_t* jobs, int njobs, void* stream);

/* ---- exact-2x bilinear up-sampling (align_corners = False) of [planes, h, w] -> [planes, 2h, 2w] and its adjoint ------
 * Replaces F.interpolate(y, size=2x, mode='bilinear', align_corners=False) in the pixel decoder's FPN path
 * (mmdet/models/layers/pixel_decoder.py:456-460).  w must be even (16-byte output stores). */
int s2f_upsample2x_fwd(const float* x, float* y, int64_t planes, int h, int w, void* stream);
/* y = sigmoid(up2x(x)) in one pass (w % 4 == 0, x and y 16-byte aligned): the inference post-processing's
 * F.interpolate(mask_pred, size = img_shape) followed by mask_pred.sigmoid() (mmseg/models/decode_heads/maskformer_head.py:170-176)
 * where the masks are predicted at half the image resolution (every Spike2Former config).  No adjoint: inference only. */
int s2f_upsample2x_sigmoid_fwd(const float* x, float* y, int64_t planes, int h, int w, void* stream);
int s2f_upsample2x_bwd(const float* gy, float* gx, int64_t planes, int h, int w, void* stream);
/* gx = adjoint(gy) + add  (add? [planes, h, w], NULL = absent): the pixel decoder's level map feeds the next level's up-sampling
 * AND the transformer decoder (pixel_decoder.py:437-449, 466-471); the gradient the second reader sends back is summed here instead
 * of by an add launch of the autograd engine. */
int s2f_upsample2x_bwd_add(const float* gy, const float* add, float* gx, int64_t planes, int h, int w, void* stream);

/* ---- batched transposition of the last two dimensions: x [B, R, C] -> y [B, C, R] (fp32) -----------------------------------
 * Replaces the `.permute(0, 1, 3, 4, 2)` / `.permute(0, 1, 4, 2, 3)` copies around the DCNv3 sampling core
 * (ops_dcnv3/modules/dcnv3.py:198-233; mmdet/models/layers/detr_layers.py:331-337).  x != y; B < 65536. */
int s2f_transpose_last2(const float* x, float* y, int64_t B, int R, int C, void* stream);
/* y = x^T + add  (add? of y's shape [B, C, R], NULL = absent): as the adjoint of a transposition whose SOURCE has a second reader --
 * the decoder layer's result goes to the prediction stack and, transposed, to the next layer (detr_layers.py:556, maskformer_head.py:554-566)
 * -- the second reader's gradient is summed here instead of by an add launch of the autograd engine. */
int s2f_transpose_last2_add(const float* x, const float* add, float* y, int64_t B, int R, int C, void* stream);
/* y[b][c][r] = q[b][c][r] + g[c] * x[b][r][c]  (x [B, R, C] token-major, q / y [B, C, R] channel-major, g [C]): the pixel decoder's
 * `query + gamma3 * ffn(query)` (detr_layers.py:336-337) with the FFN output's reinterpretation (mmcv_spike/transformer.py:829) as
 * one pass.  Backward: gx[b][r][c] = g[c] gy[b][c][r], gg[c] += sum gy[b][c][r] x[b][r][c] (gg zeroed by the caller; d/dq = gy).
 * R % 64 == 0, C % 64 == 0. */
int s2f_transpose_scale_add_fwd(const float* x, const float* q, const float* g, float* y, int64_t B, int R, int C, void* stream);
int s2f_transpose_scale_add_bwd(const float* gy, const float* x, const float* g, float* gx, float* gg_zeroed, int64_t B, int R, int C,
                                void* stream);

/* ---- mask losses of the Hungarian-matched loss on the 2x up-sampled logits -------------------
 * For matched prediction p: u = bilinear2x(pred[p]) (F.interpolate align_corners=False, dense_heads/maskformer_head.py:475-479),
 * s = sigmoid(u), t = tgt[gt_index[p]] (uint8 0/1, [2h, 2w]):  sums[p] = { sum s t, sum s, sum t, sum focal(u, t) } with the
 * sigmoid focal loss of losses/focal_loss.py:36-44; the naive dice loss (losses/dice_loss.py:45-50) follows from the first
 * three.  Nothing of size [P, 2h, 2w] is materialised forward; backward writes d(sum_k g_sums[p][k] sums[p][k])/du, to be
 * pulled back to the low-resolution logits by s2f_upsample2x_bwd.  w even; targets 4-byte aligned rows (2w % 4 == 0). */
int s2f_mask_loss_fwd(const float* pred, const uint8_t* tgt, const int64_t* gt_index, float* sums, int64_t P, int h, int w,
                      float alpha, float gamma, void* stream);
int s2f_mask_loss_bwd(const float* pred, const uint8_t* tgt, const int64_t* gt_index, const float* g_sums, float* gup, int64_t P,
                      int h, int w, float alpha, float gamma, void* stream);

/* ---- the same loss against SEMANTIC maps: disjoint targets "seg == class" (mmseg/models/decode_heads/maskformer_head.py:53-106) --
 * Matching costs (mmdet task_modules/assigners/match_cost.py:289-297 FocalLossCost(binary_input), :361-371 DiceCost) as segmented
 * sums by label instead of three [L*Q, hw] x [hw, n_gt] products: for image b, prediction row r, class id c < K
 *   out[b][r][c] = sum_{pixels of c} (pos - neg),  out[b][r][K + c] = sum_{pixels of c} s,
 *   out[b][r][2K] = sum_all neg,  out[b][r][2K + 1] = sum_all s        (s = sigmoid(pred), pos / neg the two focal terms)
 * pred [B, R, hw] fp32, seg_small [B, hw] uint8 label map at the predictions' resolution (ids >= K, e.g. the ignore label 255,
 * belong to no class), out [B, R, 2K + 2] fp32.  Sums are accumulated in 64-bit fixed point: run-to-run identical.  hw % 4 == 0.
 * Mask losses: row = (b, r) has the target  seg[b] == row_class[row]  (seg [B, 2h, 2w] uint8; row_class int32, < 0: unmatched row,
 * sums 0 / gradient 0): sums [B*R, 4] as s2f_mask_loss_fwd; the backward returns the gradient w.r.t. the LOW-resolution logits
 * [B, R, h, w] directly (the adjoint of the up-sampling is applied to an LDS tile; no [rows, 2h, 2w] tensor exists). */
int s2f_mask_cost_bins(const float* pred, const uint8_t* seg_small, float* out, int B, int R, int64_t hw, int K, float alpha,
                       float gamma, float eps, void* stream);
/* partials: workspace of s2f_mask_loss_seg_partials(B, R, h, w) floats (need not be zeroed): a row's pixels are cut into chunks whose
 * partial sums are stored there and added in chunk order -- the loss is bit-repeatable (rounds 2-4: fp32 atomics in arrival order). */
int64_t s2f_mask_loss_seg_partials(int B, int R, int h, int w);
int s2f_mask_loss_seg_fwd(const float* pred, const uint8_t* seg, const int32_t* row_class, float* sums, float* partials, int B, int R,
                          int h, int w, float alpha, float gamma, void* stream);
int s2f_mask_loss_seg_bwd(const float* pred, const uint8_t* seg, const int32_t* row_class, const float* g_sums, float* gpred, int B,
                          int R, int h, int w, float alpha, float gamma, void* stream);

/* ---- a5 / a10: spike-driven (softmax-free) attention core --------------------------------------------
 * Replaces  kv = k^T @ v ; o = (q @ kv) * scale ; o.transpose(3,4).reshape(T,B,C,N)
 * (MS_Attention_RepConv_qkv_id, mmseg/models/backbones/sdtv2.py:308-339) and the decoder's
 * (q k^T / sqrt(C)) v  ((Cross)MultiHeadAttentionBlock, mmcv_spike/transformer.py:253-274, :334-355 -- no softmax,
 * so the same bilinear form) on spikes laid out channel-major [TB, C, N] (channel c = head*d + j), i.e. directly on
 * the *_spike outputs without the reference's permute/contiguous copies.  d = C / heads <= 64.
 * q, o: [TB, C, Nq]; k, v: [TB, C, Nk]; kv_save: [TB, heads, d, d] (kept for backward).
 * With spike inputs (multiples of 1/D) every partial sum is an exactly representable fp32 while it stays below
 * 2^24 ulps -- the result is then independent of summation order and equals the reference's bit for bit. */
int s2f_sdsa_fwd(const float* q, const float* k, const float* v, float* o, float* kv_save, int TB, int heads, int d,
                 int Nq, int Nk, float scale, void* stream);
/* gq = scale * go kv^T ; gkv = scale * q^T go ; gk = v gkv^T ; gv = k gkv.   gkv_ws: [TB, heads, d, d] scratch. */
int s2f_sdsa_bwd(const float* q, const float* k, const float* v, const float* kv_save, const float* go, float* gq,
                 float* gk, float* gv, float* gkv_ws, int TB, int heads, int d, int Nq, int Nk, float scale,
                 void* stream);
/* The two building blocks (exported for tests and for callers that fuse differently):
 *   kv[tb,h][i][j]    = alpha * sum_n k[tb, h*d+i, n] * v[tb, h*d+j, n]
 *   y[tb, h*d+j, n]   = alpha * sum_i x[tb, h*d+i, n] * m[tb,h][i][j]      (transpose_m != 0: m[j][i]) */
int s2f_sdsa_kv(const float* k, const float* v, float* kv, int TB, int heads, int d, int N, float alpha, void* stream);
int s2f_sdsa_apply(const float* x, const float* m, float* y, int TB, int heads, int d, int N, float alpha,
                   int transpose_m, void* stream);

/* The same core on bf16 spike operands (uint16_t storage; what the neuron kernels write with y_bf16): exact, half the bytes.
 * q / k / v may be channel ranges of ONE tensor (the stacked q|k|v map [TB, 3C, N] of the batched projection chain): each
 * operand comes with the distance between its batch elements, in elements (multiples of 4; C*N for a plain [TB, C, N]).
 * s2f_sdsa_bwd_bf16: `go` is the gradient of o [TB, C, Nq] -- or, with go_mask != NULL, the gradient of the SPIKES
 * y = Q_IFNode(o) of the fused forward below: the straight-through estimator (in-range bit ? g / D : 0, mask layout of
 * s2f_lif_mask_words over the contiguous [TB, C, Nq]) is applied inside the loaders, no separate neuron backward pass.
 * gq / gk / gv are fp32 with their own batch strides (three ranges of one [TB, 3C, N] gradient, or three tensors). */
int s2f_sdsa_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_batch_stride,
                      int64_t k_batch_stride, int64_t v_batch_stride, float* o, float* kv_save, int TB, int heads, int d,
                      int Nq, int Nk, float scale, void* stream);
int s2f_sdsa_bwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_batch_stride,
                      int64_t k_batch_stride, int64_t v_batch_stride, const float* kv_save, const float* go,
                      const uint64_t* go_mask, int D, float* gq, float* gk, float* gv, int64_t gq_batch_stride,
                      int64_t gk_batch_stride, int64_t gv_batch_stride, float* gkv_ws, int TB, int heads, int d, int Nq,
                      int Nk, float scale, void* stream);
/* Attention core AND the neuron behind it as one kernel -- north_star's "spike-masked attention + LIF as one CDNA4 kernel"
 * for the backbone's self-attention (sdtv2.py:335-342: kv = k^T v; o = scale q kv; attn_spike(o)): one workgroup per
 * (tb, head); kv on the matrix cores straight from memory (contraction-contiguous bf16 rows, exact), kept in LDS for the
 * q kv product; o never reaches HBM: the Q_IFNode update of s2f_lif_fwd (reset membrane) runs in the epilogue and writes
 * y as bf16 spikes [TB, C, N] + the 1-bit in-range mask (mask? NULL == no backward) + the firing counters (stats? as
 * s2f_lif_fwd).  kv_save [TB, heads, d, d] leaves for the backward pass.  Needs N % 256 == 0, batch strides % 8 == 0. */
int s2f_sdsa_lif_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_batch_stride,
                          int64_t k_batch_stride, int64_t v_batch_stride, uint16_t* y_spikes, uint64_t* mask, uint64_t* stats,
                          float* kv_save, int TB, int heads, int d, int N, float scale, float vth, int D, void* stream);
/* The same without the in-range mask -- inference, where nothing is back-propagated -- for any Nq % 4 == 0 and separate key length
 * Nk: the decoder's 100-query attention blocks (mmcv_spike/transformer.py:238-300).  kv_ws [TB, heads, d, d] is scratch. */
int s2f_sdsa_lif_fwd_bf16_nomask(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_batch_stride,
                                 int64_t k_batch_stride, int64_t v_batch_stride, uint16_t* y_spikes, uint64_t* stats, float* kv_ws,
                                 int TB, int heads, int d, int Nq, int Nk, float scale, float vth, int D, void* stream);

/* ---- a9: DCNv3 core ---------------------------------------------------------------------------------
 * Replaces dcnv3_core_pytorch (ops_dcnv3/functions/dcnv3_func.py:147-189; = the dormant CUDA op
 * dcnv3_forward/backward, ops_dcnv3/src/dcnv3.h:20-59, same geometry tuple).  Layouts:
 *   input [N, H, W, G*Cg]; offset [N, Ho, Wo, G*K*K*2] ((x, y) per tap, tap k = i_w*Kh + j_h);
 *   mask [N, Ho, Wo, G*K*K]; output [N, Ho, Wo, G*Cg].   Bilinear, zero outside the zero-padded input. */
int s2f_dcnv3_fwd(const float* input, const float* offset, const float* mask, float* output, int N, int H, int W,
                  int G, int Cg, int Kh, int Kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                  float offset_scale, void* stream);
/* All three gradients are fully overwritten (grad_input need not be zeroed).  When one (n, group) slice fits in the
 * CU's LDS (12*H*W*Cg bytes + a staging chunk <= 160 KiB) the scatter-add into grad_input runs in LDS in 64-bit fixed
 * point (order-independent, see csrc/dcnv3.hip); larger maps fall back to fp32 global atomics after a memset. */
int s2f_dcnv3_bwd(const float* input, const float* offset, const float* mask, const float* grad_output,
                  float* grad_input, float* grad_offset, float* grad_mask, int N, int H, int W, int G, int Cg, int Kh,
                  int Kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, float offset_scale,
                  void* stream);

/* ---- f2: one training iteration's parameter update (csrc/optim.hip, round 5) -------------------------------------------
 * Replaces mmengine OptimWrapper.update_params for the Spike2Former configs
 * (configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:137-155): torch.nn.utils.clip_grad_norm_(max_norm = 0.01,
 * norm_type = 2) + torch.optim.AdamW.step() with one (lr, weight_decay) pair per parameter (`custom_keys` multipliers), over
 * the flat gradient buffer of the data-parallel step (dist.FlatGradAllReduce: 16-byte-aligned slots, zero pads).
 *   s2f_grad_sqnorm_parts(n)   number of fp64 partials s2f_grad_sqnorm writes for n elements
 *   s2f_grad_sqnorm            partials[i] = sum of g^2 over the i-th run of 16 384 elements (plain stores: bit-repeatable)
 *   s2f_adamw_prepare          state[0] = clip coefficient min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: 1), [1] = norm,
 *                              [4] += 1 (the step count t), [2] = 1 - beta1^t, [3] = sqrt(1 - beta2^t); state = float[8], zeroed
 *                              by the caller before the first iteration
 *   s2f_adamw_step             slots int64 [nslots][3] = {parameter pointer, slot offset in g / m / v (elements, % 4 == 0),
 *                              numel}; hyper float [nslots][2] = {lr, weight_decay} of THIS iteration; chunks int32 [nchunks][2]
 *                              = {slot, first element} covering every slot in runs of s2f_adamw_chunk_elems();
 *                              g' = g * state[0]; p *= 1 - lr wd; m += (g' - m)(1 - beta1); v = beta2 v + (1 - beta2) g'^2;
 *                              p -= lr / state[2] * m / (sqrt(v) / state[3] + eps)      (torch/optim/adamw.py, single-tensor form)
 * All tables are DEVICE memory; nothing synchronises, so the three launches can be captured behind the step's hipGraph. */
int64_t s2f_grad_sqnorm_parts(int64_t n);
int s2f_grad_sqnorm(const float* g, int64_t n, double* partials, void* stream);
int s2f_adamw_prepare(const double* partials, int nparts, float max_norm, double beta1, double beta2, float* state, void* stream);
int s2f_adamw_chunk_elems(void);
int s2f_adamw_step(const int64_t* slots, const float* hyper, const int32_t* chunks, int nchunks, const float* g, float* m, float* v,
                   const float* state, double beta1, double beta2, float eps, void* stream);

/* ---- general strided fp32 product on the vector ALUs (csrc/bmm.hip, round 6) -----------------------------------------------
 * C[b][m][n] = sum_k A[b][m][k] B[b][k][n], every operand with explicit ELEMENT strides (a transposed or broadcast operand is a
 * stride choice: *_sb = 0 shares one matrix over the batch); reduce_batch != 0: C[m][n] = sum_b sum_k ... (c_sb ignored) -- the
 * weight-gradient form.  Ascending-k (then ascending-b) fp32 multiply-adds: bit-repeatable.  Any M, N, K >= 0, any alignment.
 * Serves the shapes the matrix-core GEMM families do not take (rows that are no whole 16-byte groups, maps below one 128-column
 * tile: the plumbing configuration C1's 4 x 4 .. 16 x 16 maps and 10-query rows) -- replaces torch.bmm (rocBLAS / hipBLASLt), i.e.
 * the reference's nn.Conv2d / nn.Conv1d / nn.Linear and their autograd gradients on such shapes
 * (mmseg/models/backbones/sdtv2.py:112-255; mmdet/models/layers/transformer/mmcv_spike/transformer.py:196-361, 710-784). */
int s2f_bmm_f32(const float* a, int64_t a_sb, int64_t a_sm, int64_t a_sk, const float* b, int64_t b_sb, int64_t b_sk, int64_t b_sn,
                float* c, int64_t c_sb, int64_t c_sm, int64_t c_sn, int B, int M, int N, int K, int reduce_batch, void* stream);

/* ---- reductions and fills of the step outside the neuron / BatchNorm / GEMM kernels (csrc/glue.hip, round 6) --------------------
 * What ran as ATen reduce / fill launches inside the captured step.  Partial sums are STORED and added in index order by a second
 * small kernel: deterministic, no atomics.
 *   s2f_sum_all       out[0] = scale * sum_i x[i]; partials: float[s2f_sum_all_parts(n)] scratch.  The benchmark's headline loss
 *                     (the mean of all_cls_scores / all_mask_preds, mmdet dense_heads/maskformer_head.py:498-586 outputs).
 *   s2f_fill          p[i] = (value_ptr? ? value_ptr[0] : 1) * value -- a device scalar times a host scalar, no host round trip: the
 *                     constant gradient of a mean, written once in the layout its consumer reads.
 *   s2f_channel_sum   out[c] (+)= sum_n sum_l x[n][c][l], x [N][C][L], L % 4 == 0; workspace: float[C * s2f_channel_sum_slices(N, C, L)].
 *                     The level-embedding gradient of the decoder's fused key / value neurons (maskformer_head.py:535-540), the
 *                     mask contraction's row sums.
 *   s2f_sum_lead      out[m] = sum_t x[t][m], x [T][M], M % 4 == 0: the query position embedding's gradient, summed over the T time
 *                     steps (mmcv_spike/transformer.py:626-629). */
int64_t s2f_sum_all_parts(int64_t n);
int s2f_sum_all(const float* x, int64_t n, float scale, float* partials, float* out, void* stream);
int s2f_fill(float* p, int64_t n, const float* value_ptr, float value, void* stream);
int s2f_channel_sum_slices(int N, int C, int L);
int s2f_channel_sum(const float* x, int N, int C, int L, float* workspace, float* out, int accumulate, void* stream);
int s2f_sum_lead(const float* x, int T, int64_t M, float* out, void* stream);
/* The residual glue of a step as generic strided kernels (host: ops/glue_mode.py, a TorchDispatchMode that routes the aten calls autograd
 * and the module code still make -- gradient accumulation where two consumers meet, scalar multiples, sigmoid, layout copies, dtype
 * casts, small sums -- to these instead of ATen's kernels, and reports what it could not route).
 *   s2f_ew          out[..] = f(a[..], b?[..]) over an index space of ndim <= 6 dimensions; size / sa / sb / so: extents and ELEMENT strides
 *                   (0 = broadcast), host arrays of ndim entries.  op: 0 copy | 1 a + alpha b | 2 a b | 3 a / b | 4 sigmoid(a) |
 *                   5 (a (1 - b)) b  [sigmoid_backward(grad = a, output = b)] | 6 a alpha + beta | 7 a / alpha | 8 a - alpha b.
 *                   a_bf16: a is bf16 (a dtype cast); flat != 0: every operand contiguous over the same elements (16-byte accesses).
 *   s2f_reduce_sum  out[o] = scale * sum_r a[off(o) + off(r)]: kept index space (nd_o <= 6; may be 0) and reduced index space (1 <= nd_r <= 6),
 *                   the reduced space cut into pieces whose partial sums are stored and added in order (deterministic);
 *                   one wavefront per (output, piece), or one lane per output when consecutive outputs are adjacent in memory. */
int s2f_ew(int op, const void* a, const float* b, float* out, int ndim, const int64_t* size, const int64_t* sa, const int64_t* sb,
           const int64_t* so, float alpha, float beta, int a_bf16, int flat, void* stream);
/* up to eight contiguous fp32 pieces (each a multiple of 4 elements, 16-byte aligned) copied back to back into dst: torch.stack / cat along
 * the leading dimension as one launch; srcs / ns: HOST arrays of `count` entries */
int s2f_copy_segments(float* dst, const void* const* srcs, const int64_t* ns, int count, void* stream);
/* out = ((srcs[0] + srcs[1]) + srcs[2]) + ...  over n fp32 elements, 1 .. 16 addends (HOST array of device pointers), in the order given:
 * the gradient of a tensor with MANY readers (the decoder's query position embedding: twelve neurons per step,
 * mmcv_spike/transformer.py:597-638) in one launch and in the order the autograd engine would have accumulated it. */
int s2f_sum_n(const void* const* srcs, int count, float* out, int64_t n, void* stream);
int64_t s2f_reduce_sum_workspace(int64_t n_out, int64_t n_red);          /* floats of scratch for the call below */
int s2f_reduce_sum(const float* a, float* out, float* workspace, int nd_o, const int64_t* size_o, const int64_t* sa_o, const int64_t* so,
                   int nd_r, const int64_t* size_r, const int64_t* sa_r, float scale, void* stream);

/* ---- masked spike-driven attention (csrc/sdsa_masked.hip, round 6) -------------------------------------------------------------
 * The `attn_mask` branch of (Cross)MultiHeadAttentionBlock.forward (mmcv_spike/transformer.py:259-272, 343-355):
 *     scores = q k^T * scale ; scores.masked_fill(mask, 0) ; out = scores v
 * q, o, go, gq: fp32 [TB][heads * d][Nq]; k, v, gk, gv: fp32 [TB][heads * d][Nk] (channel-major, c = head * d + j, tb = t * B + b);
 * mask: uint8 [B][heads][Nq][Nk], non-zero = masked, shared by the time steps (the reference reshapes attn_mask with t where its comment
 * says bs and therefore only runs for t == b -- where it applies mask[b, h] to every t: the semantics implemented here, for any t).
 * Explicit O(Nq Nk d) evaluation on the vector ALUs (a mask forbids the q (k^T v) association of s2f_sdsa_*); d <= 64.  The head of
 * every shipped config passes no mask (dense_heads/maskformer_head.py:554-564). */
int s2f_sdsa_masked_fwd(const float* q, const float* k, const float* v, const uint8_t* mask, float* o, int TB, int B, int heads, int d,
                        int Nq, int Nk, float scale, void* stream);
int s2f_sdsa_masked_bwd(const float* q, const float* k, const float* v, const uint8_t* mask, const float* go, float* gq, float* gk,
                        float* gv, int TB, int B, int heads, int d, int Nq, int Nk, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* S2F_H */
