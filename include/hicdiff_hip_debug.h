/*
 * hicdiff_hip_debug.h -- test-only entry points of libhicdiff_hip.so (not part of the drop-in
 * boundary): run one convolution through the implicit-GEMM kernel, and capture intermediate
 * activations of a forward pass, so tests/ can localise a parity failure to one kernel.
 */
#ifndef HICDIFF_HIP_DEBUG_H
#define HICDIFF_HIP_DEBUG_H
#include "hicdiff_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* out[B][H][W][Cout] = conv(T(in0 ++ in1), w) + bias with the same kernel the networks use.
 *   in0/in1: NHWC device tensors (in1 may be NULL), stored size IH x IW; w: torch layout
 *   [Cout][Cin][K][K] (for unshuffle: [Cout][4*Cin][1][1]); mode bits: 1 = nearest x2 upsample,
 *   2 = weight-standardise, 4 = pixel-unshuffle 2x2 stride 2, 8 = affine+SiLU input transform with
 *   A,Bv [B][Cin] (E optional), 16 = LayerNorm input transform (stats computed internally, g = A),
 *   32 = split-bf16 x3 arithmetic instead of exact fp32, 64 = use the 32-channel K slice, 128 = LayerNorm statistics passed in Bv,
 *   256 (with 32; 3x3 only) = also pack the Winograd F(2x2,3x3) filter image, so the convolution runs on conv_winograd.hip -- an
 *   error if the shape is not one that kernel takes; 512 (with 32; 3x3 only) = two fp16 products per multiply, xh (wh + wl); 1024 (with 512) = one, xh wh;
 *   2048 (with 32 | 64) = ONE bf16 product per multiply (the training step's optional arithmetic; 32-channel slices, plain / GroupNorm-apply loaders --
 *   three products where that kernel form does not exist). */
int hd_debug_conv(const float* in0, int C0, const float* in1, int C1, int B, int IH, int IW, const float* w,
                  const float* bias, int Cout, int K, int mode, const float* A, const float* Bv, const float* E,
                  float* out, void* stream);

/* The fused q side of LinearAttention (src/hicdiff.py:217-226,207-210): out = LayerNorm(to_out(einsum(context,
 * softmax_d(q) * scale))) * g + res, run as the engine runs it: the context is folded into to_out's weight per
 * sample, the softmax is the conv's loader transform, LayerNorm + residual its epilogue (C = 64 / 128) or a second
 * kernel.  q: NHWC [B,H,W,128] (4 heads x 32); ctx: [B,4,32,32] (d, e); wout: torch layout [C,128,1,1]; H*W >= 256. */
int hd_debug_linattn_out(const float* q, const float* ctx, const float* wout, const float* bias, const float* g, const float* res,
                         int B, int H, int W, int C, float* out, void* stream);

/* The chained q side for 64-channel maps (linattn_q_fused.hip): out = LayerNorm(to_out(einsum(context,
 * softmax_d(to_q(LayerNorm(x) * norm_g)) * scale))) * gout + x in one kernel.  x, out: NHWC [B,H,W,64];
 * wqkv: torch layout [384,64,1,1] (rows 0..127 are q); ctx: [B,4,32,32]; wout: [64,128,1,1]. */
int hd_debug_linattn_q(const float* x, const float* norm_g, const float* wqkv, const float* ctx, const float* wout, const float* bias,
                       const float* gout, int B, int H, int W, float* out, void* stream);

/* The weight-gradient component of the training step (csrc/train.hip, struct Wgrad) on an arbitrary shape:
 * dW[Cout][C0+C1][KT][KT] = d/dW of conv2d(cat(x0, x1), W, padding = KT/2) contracted with g -- torch.nn.grad.conv2d_weight.
 * x0: NHWC [B,H,W,C0]; x1: NHWC [B,H,W,C1] or NULL; g: NHWC [B,H,W,Cout]; KT = 3 or 1; W <= 64; Cout % 64 == 0.
 * affA / affB ([B][C0+C1], may be NULL): the input is silu(x * affA + affB) per (sample, channel) -- a normalised activation
 * recomputed on the way in.  plain bit 0: one bf16 product instead of three; plain >> 1 = source addressing of x0: 1 = x0 is the half-size map
 * [B,H/2,W/2,C0] taken through a nearest x2 upsample (Upsample, src/hicdiff.py:72-76), 2 = x0 is the double-size map [B,2H,2W,C0/4] taken through
 * the pixel-unshuffle (Downsample, :78-82; channel c*4 + p1*2 + p2).  Allocates and frees its operand images; synchronises. */
int hd_debug_conv_wgrad(const float* x0, int C0, const float* x1, int C1, const float* g, int B, int H, int W, int Cout, int KT,
                        const float* affA, const float* affB, int plain, float* dW, void* stream);
/* The same gradient straight from the NHWC tensors (csrc/train.hip, wgrad_direct_kernel: no operand rewrite; split-bf16 x3 only), plus the
 * bias gradient db[Cout] = sum over pixels of g (db may be NULL).  affA / affB / affE ([B][C0+C1], may be NULL; KT = 3 only): the input is
 * silu(x * affA + affB) + affE per (sample, channel).  src_mode: source addressing of x0 as hd_debug_conv_wgrad's plain >> 1 (0, 1 = upsample, 2 = unshuffle). */
int hd_debug_conv_wgrad_direct(const float* x0, int C0, const float* x1, int C1, const float* g, int B, int H, int W, int Cout, int KT,
                               float* dW, float* db, const float* affA, const float* affB, const float* affE, int src_mode, void* stream);

/* Backward components of the UNet's normalisations (csrc/train_norms.hip; groundwork for its training step), against torch autograd:
 *   hd_debug_gn_silu_bwd  y = silu(GroupNorm_G(x) * (scale + 1) + shift) (src/hicdiff.py:155-171): x NHWC [B,H,W,C]; g holds dL/dy on entry and
 *                         dL/dx on return; film = [B][2C] (scale | shift) or NULL (no FiLM); dgamma, dbeta [C]; dfilm [B][2C] (may be NULL).
 *   hd_debug_ln_bwd       y = (x - mean_c) rsqrt(var_c + 1e-5) * gain per pixel row (src/hicdiff.py:99-108): dy -> dx in place, dgain [C].
 *   hd_debug_ws_bwd       dW from d(standardised W) per output filter (src/hicdiff.py:84-97); n = Cin*KH*KW.
 * They allocate their scratch, run on the stream and synchronise. */
int hd_debug_gn_silu_bwd(const float* x, float* g, const float* gamma, const float* beta, const float* film, int B, int H, int W, int C, int G,
                         float* dgamma, float* dbeta, float* dfilm, void* stream);
int hd_debug_ln_bwd(const float* x, float* dy, const float* gain, long long P, int C, float* dgain, void* stream);
int hd_debug_ws_bwd(const float* w, const float* dwhat, int Cout, int n, float* dw, void* stream);

/* First / last convolutions of the UNet: hd_debug_first_conv_wgrad: dW[C][J][KS][KS] of a KS x KS (3 or 7), same-padded convolution of J (1 or 2)
 * single-channel planes in0, in1 ([B][S][S]) into C channels, given the gradient g ([B,S,S,C]) of its output (init_conv, src/hicdiff.py:279).
 * hd_debug_rowdot_bwd: the 1x1 convolution C -> 1 (final_conv, :319): dx[p][c] = dout[p] w[c], dw[c] = sum_p x[p][c] dout[p]. */
int hd_debug_first_conv_wgrad(const float* g, const float* in0, const float* in1, int J, int B, int S, int C, int KS, float* dW, void* stream);
int hd_debug_rowdot_bwd(const float* x, const float* dout, const float* w, long long P, int C, float* dx, float* dw, void* stream);

/* Gradient stages (hd_train_stage_*): with a snapshot buffer set (float[total params], device; NULL = off) every hd_train_loss_backward copies a
 * stage's slots into it in stream order right behind that stage's event.  The snapshot equals the step's final gradients iff no kernel writes
 * a slot after its stage's event -- the property an all-reduce that starts at the event relies on. */
int hd_debug_train_stage_snapshot(hd_trainer* t, float* snapshot);

/* Gradient routing of the resampling layers: which == 1: 2x2 sum-pool (g [B,2H,2W,C] -> dx [B,H,W,C]); which == 2: pixel-shuffle (g [B,H,W,4C] ->
 * dx [B,2H,2W,C], g's channel c*4 + p1*2 + p2 goes to pixel (2y+p1, 2x+p2)).  C here is dx's channel count. */
int hd_debug_resample_bwd(const float* g, int B, int H, int W, int C, int which, float* dx, void* stream);

/* Backward of the attention cores (csrc/train_attn.hip), against torch autograd.  qkv: NHWC [B][n][3*heads*32] (q | k | v); dout: [B][n][heads*32];
 * dqkv like qkv.  hd_debug_attn_full_bwd: softmax attention of the mid block (src/hicdiff.py:239-251), n <= 64 tokens. */
int hd_debug_attn_full_bwd(const float* qkv, const float* dout, int B, int n, int heads, float* dqkv, void* stream);
/* hd_debug_linattn_bwd: the LinearAttention core (src/hicdiff.py:212-224: q softmax over d, k softmax over tokens, q * scale, v / n, context, out), any n. */
int hd_debug_linattn_bwd(const float* qkv, const float* dout, int B, int n, int heads, float* dqkv, void* stream);

/* The Winograd F(2x2,3x3) form of the eligible 3x3 convolutions (csrc/conv_winograd.hip) is opt-in: 1 = on, 0 = off, -1 = as the
 * environment says (HICDIFF_WINOGRAD, default off).  Process-wide; a context packs the Winograd filter images at hd_load_weights only while
 * the switch is on, so turn it on BEFORE loading the weights (layers without the image take the implicit-GEMM kernel). */
int hd_debug_winograd(int mode);

/* Enable capture (1) / disable and drop captures (0) of labelled intermediates of later forwards. */
int hd_debug_capture(hd_ctx* ctx, int enable);

/* The device Gaussian generator exactly as the fused sampler steps draw from it: Philox4x32-10 keyed by seed, counter = (pixel quad,
 * global tile index, step, noise_stream), Box-Muller.  hd_randn is noise_stream 0 (the ancestral step's z and DDRM's 'missing' draw,
 * src/functions/denoising.py:92); 1 = DDRM's 'after' draw (:96), 2 = its 'before' draw (:100).  Tests read the chain's noise with this
 * and replay it through the CPU oracle. */
int hd_debug_randn(float* out, int B, int S, uint64_t seed, uint64_t tile_offset, uint32_t step, uint32_t noise_stream, void* stream);
/* Copy capture `label` (NHWC fp32) to the DEVICE buffer dst (capacity n floats); dims = {B,H,W,C}.
 * Returns HD_EINVAL when the label was not captured. */
int hd_debug_read(hd_ctx* ctx, const char* label, float* dst, size_t n, int32_t dims[4]);

#ifdef __cplusplus
}
#endif
#endif
