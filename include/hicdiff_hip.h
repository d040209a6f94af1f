/*
 * hicdiff_hip.h -- C ABI of libhicdiff_hip.so: the MI355X (gfx950) implementation of HiCDiff's
 * per-timestep hot path (epsilon-network forward + DDPM / DDRM update) over 1 x S x S Hi-C tiles.
 *
 * The reference (BioinfoMachineLearning/hicdiff) is pure Python/PyTorch and has no FFI layer; the
 * boundary it offers is its Python object contract.  Each entry point below
 * names the reference call it stands in for (paths relative to the upstream checkout).  The host
 * side that mirrors the reference classes (hicdiff_amd/) binds these symbols with ctypes; a
 * maintainer of the reference would add the same binding.
 *
 * Conventions
 *   - every function returns 0 on success or a negative HD_E* code; no C++ exception crosses the
 *     ABI; hd_last_error() gives the message of the last failure on that context;
 *   - all tensor pointers are DEVICE pointers to contiguous fp32 unless stated otherwise; tiles
 *     are (B, 1, S, S) row-major exactly as the reference's torch tensors;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it: no host
 *     synchronisation, no allocation (workspace is sized by hd_reserve, called outside capture);
 *   - one host thread drives one context; contexts are independent (one per GPU / per rank).
 */
#ifndef HICDIFF_HIP_H
#define HICDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hd_ctx hd_ctx;

enum {
    HD_OK = 0,
    HD_EINVAL = -1,     /* bad argument / unsupported architecture parameter */
    HD_ENOWEIGHT = -2,  /* a required state-dict entry is missing or has the wrong shape */
    HD_EHIP = -3,       /* a HIP runtime call failed */
    HD_ENOMEM = -4,     /* workspace too small: call hd_reserve(ctx, B) first */
    HD_ESTATE = -5      /* weights not loaded yet */
};

enum { HD_ARCH_UNET = 0, HD_ARCH_HICEDRN = 1 };

/* dtype of the per-sample time argument of the epsilon-network */
enum {
    HD_T_INT64 = 0,     /* torch.long timesteps, GaussianDiffusion.p_sample (src/hicdiff.py:597) */
    HD_T_FLOAT32 = 1    /* float timesteps (DDRM, src/functions/denoising.py:49) or SR3 noise level
                           (B,1) (src/hicdiff_sr3.py:636) */
};

/* Architecture of the epsilon-network.  Replaces the constructor arguments of
 *   Unet(dim, dim_mults, channels, self_condition[, noise_level_emb])   src/hicdiff.py:256-269, src/hicdiff_sr3.py:311-324
 *   hicedrn_Diff(channels, number_resnet, self_condition[, noise_level_emb]) src/model/hicedrn_Diff.py:211-218 */
typedef struct {
    int32_t kind;            /* HD_ARCH_UNET | HD_ARCH_HICEDRN */
    int32_t dim;             /* UNet base width (64; multiple of 16) / hicedrn n_feat (256) */
    int32_t n_mults;         /* UNet: len(dim_mults) (<= 8) */
    int32_t mults[8];        /* UNet: dim_mults */
    int32_t channels;        /* image channels; only 1 is supported (Hi-C tiles) */
    int32_t self_condition;  /* 1: network input is cat(cond, x) on the channel axis */
    int32_t sr3;             /* 1: SR3 flavour: noise-level PositionalEncoding + additive FeatureWiseAffine */
    int32_t groups;          /* UNet GroupNorm groups (8) */
    int32_t number_resnet;   /* hicedrn residual blocks (32) */
    int32_t reserved[4];
} hd_arch_desc;

/* One state-dict entry, named exactly as in the reference checkpoint
 * (e.g. "downs.0.0.block1.proj.weight"; no "model." prefix). `data` is a device pointer to fp32
 * in the reference (torch) layout; it is only read during hd_load_weights. */
typedef struct {
    const char* name;
    const void* data;
    int32_t ndim;
    int64_t shape[4];
} hd_named_tensor;

/* ABI revision of this header.  hd_abi_version() returns the revision the library was built from: a binding checks it once after
 * loading.  Revision 3 (round 3) put `struct_bytes` in front of the two per-step coefficient structs.  Entry points
 * added since (round 4: hd_chain_begin / hd_chain_end / hd_set_chains / hd_chains_for) leave the revision alone: nothing a revision-3 binding calls
 * changed, and a binding that wants the new symbols fails at symbol lookup on an older library.
 *
 * Size-prefixed structs.  The caller stores sizeof(its own struct) in `struct_bytes`; the library copies that many bytes (never more)
 * into a zeroed struct of its own revision, so a caller built against an OLDER revision of a struct keeps working (the fields it does
 * not know read as 0) and a struct from a binding that predates the prefix, or any other size the library does not know, is refused
 * with HD_EINVAL instead of being read past its end. */
#define HD_ABI_VERSION 3

/* Coefficients of one ancestral reverse step, gathered by the host from the 13 schedule buffers of
 * GaussianDiffusion.__init__ (src/hicdiff.py:494-522) at index t. */
typedef struct {
    uint32_t struct_bytes;               /* = sizeof(hd_ddpm_coef) of the caller's header (36 with `arith`, round 4; 32 and 28 -- revision-3 structs without
                                            `arith` / without `eps_coef` -- are accepted: the missing fields read as 0) */
    float sqrt_recip_alphas_cumprod;     /* predict_start_from_noise, src/hicdiff.py:529-533 */
    float sqrt_recipm1_alphas_cumprod;
    float posterior_mean_coef1;          /* q_posterior, src/hicdiff.py:553-560 */
    float posterior_mean_coef2;
    float sigma;                         /* exp(0.5 * posterior_log_variance_clipped[t]); 0 at t == 0 */
    float time_value;                    /* value fed to the time embedding: t, or the SR3 noise level */
    float eps_coef;                      /* weight of eps itself in the update; 0 for the ancestral step.  A DDIM step (src/hicdiff.py:622-664) is
                                            coef1 = sqrt(alpha_next), coef2 = 0, eps_coef = sqrt(1 - alpha_next - sigma^2), sigma = eta-term */
    uint32_t arith;                      /* arithmetic of THIS step's 3x3 convolutions: HD_ARITH_DEFAULT (the context's, hd_set_precision) or
                                            HD_ARITH_F16W2 / HD_ARITH_F16W1 (two / one fp16 products per multiply, xh (wh + wl) / xh wh; only on a
                                            context in HD_PRECISION_BF16X3).  The host's
                                            precision schedule over a chain: hicdiff_amd/_diffusion.py, DESIGN.md section 4e.
                                            A layer whose largest packed weight lies outside [2^-8, 2^15] (measured by hd_load_weights)
                                            keeps three bf16 products whatever is asked: its fp16 hi | lo image would not be faithful. */
} hd_ddpm_coef;
enum { HD_ARITH_DEFAULT = 0, HD_ARITH_F16W2 = 1, HD_ARITH_F16W1 = 2,
       HD_ARITH_F16W2_LOW = 3 /* two fp16 products on the feature maps of at most (S/4)^2 pixels, three elsewhere */ };

/* Coefficients of one DDRM 'deno' step (src/functions/denoising.py:48-104 with identity H). */
typedef struct {
    uint32_t struct_bytes;  /* = sizeof(hd_ddrm_coef) (44 with `skip_network`, round 4; the 40 bytes of revision 3 are accepted) */
    float sqrt_at;          /* sqrt(alpha_bar_t) */
    float sqrt_1m_at;       /* sqrt(1 - alpha_bar_t) */
    float sqrt_at_next;     /* sqrt(alpha_bar_next) */
    float sigma_next;       /* sqrt(1 - a_next) / sqrt(a_next) */
    float sigma_0, etaA, etaB, etaC;
    float time_value;       /* t as float */
    uint32_t skip_network;  /* 1: do not evaluate the epsilon-network -- allowed only on a step whose update provably ignores it: etaB == 1 and
                               sigma_next > sigma_0, where every pixel takes the third case and x_next = sqrt(a_next) (y + sqrt(sigma_next^2 -
                               sigma_0^2) z), src/functions/denoising.py:99-100.  Any other step is refused with HD_EINVAL.  x_next is
                               bit-identical to the full step's; x0_out (the network's x0 estimate) is NOT produced on such a step. */
} hd_ddrm_coef;

/* ---- lifetime ------------------------------------------------------------------------------ */

/* nn.Module construction (Unet.__init__ / hicedrn_Diff.__init__). Selects `device`. */
int hd_create(hd_ctx** out, int device, const hd_arch_desc* arch);
void hd_destroy(hd_ctx* ctx);
const char* hd_last_error(const hd_ctx* ctx);
const char* hd_version(void);
int hd_abi_version(void);   /* HD_ABI_VERSION of the build */

/* load_state_dict (train.py:186, inference.py:93,104): validates names/shapes against the
 * architecture, weight-standardises the UNet's 3x3 convs once (src/hicdiff.py:84-97 recomputes it
 * every forward) and re-packs everything into kernel layout. May be called again after an
 * optimiser step. Enqueued on `stream`. */
int hd_load_weights(hd_ctx* ctx, const hd_named_tensor* tensors, int n, void* stream);

/* Workspace for batches of up to B tiles of S x S (grow-only; synchronises the device when it has
 * to reallocate, so call it outside stream capture). hd_workspace_bytes reports the size. */
int hd_reserve(hd_ctx* ctx, int B, int S);
int hd_workspace_bytes(const hd_ctx* ctx, int B, int S, size_t* out);

/* Arithmetic of the wide convolutions (every other kernel is fp32 throughout):
 *   HD_PRECISION_F32     exact fp32 on v_mfma_f32_32x32x2_f32 (strict-parity mode, 157 TFLOP/s peak);
 *   HD_PRECISION_BF16X3  default: operands split into two bf16 (x = hi + lo), three bf16 MFMAs per product
 *                        with fp32 accumulation -- ~2e-5 relative error per forward, inside the 1e-3
 *                        parity bound, at up to 5x the fp32 MFMA rate.
 *   HD_PRECISION_F16W2   BF16X3 with the 3x3 convolutions on TWO fp16 products per multiply, xh (wh + wl): the activation rounded
 *                        once to fp16, the weight exact -- ~1e-3 per forward, i.e. NOT inside the parity bound as a whole-chain mode;
 *                        it exists for the early half of a long ancestral chain, where the chain damps the error (the per-step
 *                        switch hd_ddpm_coef.arith is what the samplers use), and as a context-wide mode for tests and measurements.
 *   HD_PRECISION_F16W1   the same with ONE fp16 product, xh wh (weights rounded to fp16 as well): the earliest quarter of a long chain.
 * The environment variable HICDIFF_PRECISION=f32|bf16x3 sets the default at hd_create. */
enum { HD_PRECISION_F32 = 0, HD_PRECISION_BF16X3 = 1, HD_PRECISION_F16W2 = 2, HD_PRECISION_F16W1 = 3 };
int hd_set_precision(hd_ctx* ctx, int mode);

/* ---- DDRM with a general degradation H = U S V^T (src/functions/denoising.py:11-111 over the operators of
 * src/functions/svd_replacement.py:72-541; HiCDiff itself selects the identity, which hd_ddrm_step fuses).  All tensors are
 * device fp32; vectors are [B][D] (spectral / pixel space, D = channels * S * S, a multiple of 4) or [B][M] (measurement
 * space, M singular values). */

/* x0_t = (x_t - eps * sqrt(1 - a_t)) / sqrt(a_t)  (:66); n elements, a multiple of 4. */
int hd_ddrm_x0(const float* x, const float* eps, float sqrt_at, float sqrt_1m_at, float* x0_out, size_t n, void* stream);

/* The three-case spectral update of one step (:69-104) given V^T x0_t, V^T eps, U^T y and the singular values; `out` is
 * sqrt(a_next) * (V^T x)_next, to which the caller applies V.  c: sigma_next, sigma_0, etaA/B/C, sqrt_at_next are read.
 * z0 / z1 / z2: replayed N(0,1) in FULL layout ([B][D], [B][D], [B][M]: the draws of :92, :96 scattered to their elements, :100)
 * or NULL for device Philox (noise streams 0 / 1 / 2, key (seed, tile_offset + b, step)). */
int hd_ddrm_general_update(const float* vt_x0, const float* vt_et, const float* ut_y, const float* singulars, int M,
                           const float* z0, const float* z1, const float* z2, const hd_ddrm_coef* c, float* out, int B, int D,
                           uint64_t seed, uint64_t tile_offset, uint32_t step, void* stream);

/* Building blocks of the SVD-free operators: dst[b][i] = idx[i] >= 0 ? src[b][idx[i]] : 0 (permutations, selections, zero
 * padding); dst[n][:] = mat (K x K, row-major, K <= 64) applied to N contiguous K-vectors (src != dst); in-place fast
 * Walsh-Hadamard transform of N rows of length L = 2^p <= 4096 times `scale`. */
int hd_gather_cols(const float* src, const int* idx, float* dst, int B, int Dsrc, int Ddst, void* stream);
int hd_kvec_matmul(const float* src, const float* mat, float* dst, size_t N, int K, void* stream);
/* dst[i] = A x[i] Bm for n row-major S x S images, S <= 64 (the separable blur operators' V / V^T / U / U^T,
 * src/functions/svd_replacement.py:401-541); dst[n][m] = sum_k src[n][k] mat[k][m] for any sizes (GeneralH's dense factors, :72-107). */
int hd_sandwich_matmul(const float* A, const float* x, const float* Bm, float* dst, int n, int S, void* stream);
int hd_dense_matmul(const float* src, const float* mat, float* dst, int N, int K, int M, void* stream);
int hd_fwht(float* data, int N, int L, float scale, void* stream);

/* hd_ddpm_step / hd_ddrm_step with device-generated noise replay a captured hipGraph of the whole step
 * (one per (B, S, tensor addresses); the per-step scalars travel through a 1-thread kernel) on a
 * stream owned by the context, ordered after/before the caller's stream by events.  0 disables it
 * (every kernel is then launched eagerly on the caller's stream), 1 forces it; untouched, a context replays steps of at least
 * 150 k pixels (B*S*S) and launches smaller ones eagerly (measured faster there); env HICDIFF_GRAPHS=0|1 forces either at hd_create. */
int hd_set_graphs(hd_ctx* ctx, int enable);

/* The loop of p_sample_loop (src/hicdiff.py:603-620; conditional src/hicdiff_condition.py:676-678; the DDRM loop
 * src/functions/denoising.py:38-109) as a bracket around the per-step calls.  Between hd_chain_begin and hd_chain_end the replayed
 * steps (hd_ddpm_step / hd_ddrm_step with device noise) are ordered against `stream` only at the first step and at hd_chain_end: the
 * caller must not touch x / cond / y / x0_out on its stream inside the bracket.  A step that has to run on the caller's stream
 * (replayed noise, a batch too small to replay) joins first, so mixing stays correct, only slower.  Large steps are cut into two
 * half-batch chains (tiles are independent for the whole chain) that advance side by side on two streams of the
 * context and only meet at hd_chain_end -- results are bit-identical to the single chain.  hd_set_chains(n): n = 1 never cuts,
 * n = 2..4 always cuts into n sub-batches, 0 restores the default (two chains for every replayed step, i.e. from 150 k pixels per step on); env HICDIFF_CHAINS=n sets it at
 * hd_create.  hd_reserve / hd_set_chains inside a bracket return HD_ESTATE. */
int hd_chain_begin(hd_ctx* ctx, void* stream);
int hd_chain_end(hd_ctx* ctx, void* stream);
int hd_set_chains(hd_ctx* ctx, int n);
int hd_chains_for(const hd_ctx* ctx, int B, int S);   /* how many chains a replayed step of B tiles of S x S is cut into (1..4) */

/* ---- the hot path -------------------------------------------------------------------------- */

/* eps = model(x, t, x_self_cond): Unet.forward src/hicdiff.py:345-387, hicedrn_Diff.forward
 * src/model/hicedrn_Diff.py:267-289; called from model_predictions (src/hicdiff.py:563), p_losses
 * (:731) and the DDRM loop (src/functions/denoising.py:57).
 *   x, eps: (B,1,S,S); cond: (B,1,S,S) or NULL (must be non-NULL iff self_condition);
 *   t: B elements of t_kind. */
int hd_eps_forward(hd_ctx* ctx, const float* x, const void* t, int t_kind, const float* cond,
                   float* eps, int B, int S, void* stream);

/* One fused ancestral step, GaussianDiffusion.p_sample (src/hicdiff.py:594-601; conditional
 * src/hicdiff_condition.py:592-598; SR3 src/hicdiff_sr3.py:634-652):
 *   eps = model(x, t, cond); x0 = clamp(c.recip * x - c.recipm1 * eps, -1, 1);
 *   x <- c.coef1 * x0 + c.coef2 * x + c.eps_coef * eps + c.sigma * noise.
 * noise: (B,1,S,S) host-replayed N(0,1) for parity runs, or NULL to draw it on the device
 * (Philox4x32-10 keyed by (seed, tile_offset + tile, step)); x0_out optional (may be NULL). */
int hd_ddpm_step(hd_ctx* ctx, float* x_inout, const float* cond, const float* noise,
                 const hd_ddpm_coef* c, float* x0_out, int B, int S,
                 uint64_t seed, uint64_t tile_offset, uint32_t step, void* stream);

/* One fused DDRM denoising step (efficient_generalized_steps, src/functions/denoising.py:48-104,
 * Denoising H: src/functions/svd_replacement.py:148-168):
 *   eps = model(x, t); x0_t = (x - eps*sqrt(1-a_t))/sqrt(a_t); three-case update against y.
 * z: 3 x (B,1,S,S) replayed noise (missing / after / before draws, in that order) or NULL for
 * device Philox. x0_out optional. */
int hd_ddrm_step(hd_ctx* ctx, float* x_inout, const float* y, const float* z,
                 const hd_ddrm_coef* c, float* x0_out, int B, int S,
                 uint64_t seed, uint64_t tile_offset, uint32_t step, void* stream);

/* q_sample (src/hicdiff.py:694-700): out = a[b]*x0 + s[b]*noise with per-sample fp32 coefficients
 * already gathered by the host side (device pointers, B each). */
int hd_q_sample(hd_ctx* ctx, const float* x0, const float* noise, const float* a, const float* s,
                float* out, int B, int S, void* stream);

/* Per-sample loss of p_losses (src/hicdiff.py:743-744): out[b] = mean_pixels(|pred-target|) (l1)
 * or mean((pred-target)^2) (l2). */
int hd_loss_per_sample(hd_ctx* ctx, const float* pred, const float* target, int l2, float* out,
                       int B, int S, void* stream);

/* Fill out (B,1,S,S) with N(0,1) from the device generator (torch.randn stand-in for perf runs,
 * src/hicdiff.py:607). */
int hd_randn(hd_ctx* ctx, float* out, int B, int S, uint64_t seed, uint64_t tile_offset,
             uint32_t step, void* stream);

/* ---- evaluation (the callers next to the path) ---------------------- */

/* Tile-quality sums of the reference's evaluation loop: SSIM (src/Utils/loss/SSIM.py:17-37: 11x11 Gaussian window,
 * sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2) and the raw moments behind mse / psnr / snr / pcc
 * (src/Utils/stard_metrics.py:146-160).  pred, target: (B,1,S,S) device tensors, S <= 128.  rescale != 0 first maps
 * [-1,1] tiles to [0,1] with a clamp (inverse_data_transform('rescaled'), src/datasets/__init__.py:214-223);
 * rescale == 0 takes the values as they are (the ssim(img1, img2) call itself).
 *   sums (device, 8 doubles): sum (p-t)^2, sum of the SSIM map, sum t, sum p, sum t^2, sum p^2, sum p*t, B*S*S
 *   ssim_each (device, B floats, may be NULL): per-tile mean of the SSIM map (size_average=False)
 *   partial: device scratch, B*8 doubles.  No context needed; only enqueues on the stream; deterministic. */
int hd_tile_metrics(const float* pred, const float* target, int B, int S, int rescale, double* partial, double* sums, float* ssim_each,
                    void* stream);

/* ---- tile producer / stitcher (the data format either side of the path) ---- */

/* Cut the band of piece x piece tiles out of one chromosome's dense contact matrix: what splitPieces
 * (processdata/PrepareData_linear_sing.py:25-46) does with numpy slices.
 *   mat      device f32 [n][n], the matrix BEFORE padding; elements past its edge read as 0, which is the
 *            reference's F.pad to a multiple of piece (:33-38)
 *   origins  device i32 [ntiles][2], (row, col) of each tile's first element, in output order
 *            (hicdiff_amd.processdata.tile_origins restates the reference's double loop and band rule :40-43)
 *   tiles    device f32 [ntiles][piece][piece] (the (ntiles,1,piece,piece) array of Splits/..._full_chr_*.npy)
 * Only enqueues on the stream; bit-exact (a copy). */
int hd_split_pieces(const float* mat, int n, const int* origins, int ntiles, int piece, float* tiles, void* stream);

/* The inverse the reference lacks (its evaluation stays on tiles): write sampled tiles back into a dense n x n
 * matrix.  tile_of is a device i32 [nb][nb] table over the step grid: the index of the tile whose origin is
 * (kr*step, kc*step), or -1.  Element (r,c) takes the tile element that holds it; else the element that holds (c,r)
 * (Hi-C maps are symmetric and only upper-triangle tiles are cut); else 0.  step >= piece (tiles do not overlap:
 * the reference's cut fails for step < piece).  Only enqueues on the stream; bit-exact (a copy). */
int hd_stitch_pieces(const float* tiles, const int* tile_of, int nb, int piece, int step, float* mat, int n, void* stream);

/* ---- training step ------------------------------------------------------ */

/* Native training of the hicedrn eps-network: what `loss = diffusion(x); loss.backward(); optimizer.step()` does in
 * train.py:109-134 with torch autograd (p_losses src/hicdiff.py:711-747, src/hicdiff_condition.py:715-746; net
 * src/model/hicedrn_Diff.py:169-289; optim.Adam(lr=2e-5) train.py:111).  The caller owns four flat fp32 device arrays
 * (params, grads, and Adam's m, v) laid out as hd_train_param_slot describes: the reference's state_dict order, every tensor
 * in torch layout at a 16-byte aligned offset.  Covers both eps-networks in their three flavours: hicedrn_Diff (src/model/hicedrn_Diff.py,
 * hicedrn_sr3_Diff.py) and the UNet (src/hicdiff.py, hicdiff_condition.py, hicdiff_sr3.py; width a multiple of 64, at most four levels). */
typedef struct hd_trainer hd_trainer;

/* Sizes the saved activations for batches of exactly B tiles of 1xSxS (hicedrn: three tensors of B*S*S*256 floats per residual block -- 25.8 GB at 64 tiles of 64x64 and 32 blocks,
 * S a multiple of 8, 8..64; UNet: one arena sized by walking a step without launching, S divisible by 2^(levels-1) with a last map of at
 * least 4x4).  HD_EINVAL for shapes outside that, HD_ENOMEM when the device cannot hold them. */
int hd_train_create(hd_trainer** out, int device, const hd_arch_desc* arch, int B, int S);
void hd_train_destroy(hd_trainer* t);
const char* hd_train_last_error(const hd_trainer* t);

/* Arithmetic of the wide 3x3 convolutions (the 8-wave kernel: hicedrn's 256-channel layers, the UNet's 32 x 32 level) and of the weight
 * gradients of 3x3 layers with more than 64 input channels: HD_PRECISION_BF16X3 (default: three bf16 MFMA products per fp32 product, the
 * arithmetic of the parity tests) or HD_TRAIN_PREC_BF16 (one bf16 product, fp32 accumulate: the "bf16 compute, fp32 master weights" of
 * mixed-precision training; gradients then carry bf16 rounding, ~1e-2 relative; layers without a one-product kernel form keep the three
 * products).  Master weights, gradients, Adam state, FiLM / time MLP and every elementwise step stay fp32 in both. */
#define HD_TRAIN_PREC_BF16 2
int hd_train_set_precision(hd_trainer* t, int mode);

/* What the network's output is compared with (GaussianDiffusion(objective = ...), src/hicdiff.py:441,733-741): 0 the noise (default; the
 * only one a reference driver uses), 1 x_start, 2 v = a_t noise - s_t x_start (predict_v, :542-546).  Not for SR3 nets. */
int hd_train_set_objective(hd_trainer* t, int objective);
/* Per-sample loss weights of the following hd_train_loss_backward calls: w[b] = p2_loss_weight[t_b] (src/hicdiff.py:522,746; device
 * pointer to B floats the caller keeps alive, or NULL for weights of 1 -- the reference's own setting, p2_loss_weight_gamma = 0). */
int hd_train_set_loss_weights(hd_trainer* t, const float* w);

/* Number of parameter tensors; *total_floats = length of the flat arrays. */
int hd_train_param_count(const hd_trainer* t, long long* total_floats);
/* Slot i: state_dict key, offset in floats, shape (padded with 1s to 4 entries), rank. */
int hd_train_param_slot(const hd_trainer* t, int i, const char** name, long long* offset, long long* shape4, int* ndim);

/* x_t = a_t[b] x_start + s_t[b] noise; eps_hat = net(x_t, t, cond); loss = mean_b mean_pixels f(eps_hat - noise), f = square
 * (l2 != 0) or abs; grads <- d loss / d params (every slot overwritten).  timesteps: int64[B] (t_kind HD_T_INT64) with a_t, s_t:
 * float[B] = sqrt_alphas_cumprod[t], sqrt_one_minus_alphas_cumprod[t] (src/hicdiff.py:694-700); SR3 nets instead take the
 * continuous noise level float[B] (t_kind HD_T_FLOAT32) with a_t = level, s_t = sqrt(1 - level^2) (src/hicdiff_sr3.py:735-792).
 * cond: the low-coverage tiles iff self_condition; loss: one device float.  Only enqueues on the stream; deterministic. */
int hd_train_loss_backward(hd_trainer* t, const float* params, float* grads, const float* x_start, const float* cond,
                           const void* timesteps, int t_kind, const float* noise, const float* a_t, const float* s_t, int l2, float* loss,
                           void* stream);

/* Gradient stages: overlapping the data-parallel all-reduce with the backward pass (train.py:131-134 under torchrun).
 * hd_train_loss_backward produces the gradients back to front; the trainer cuts its parameter slots into a few stages in the order their
 * gradients become FINAL (stage 0 first: the output end of the network; the last stage: the time MLP, the FiLM projections and the first
 * convolution, which are finished at the very end) and records one event per stage on the step's stream behind the last kernel that writes
 * a gradient of that stage.  A caller reduces stage k while the kernels of the later stages still run:
 *   hd_train_stage_count   number of stages (>= 1)
 *   hd_train_slot_stage    stage of parameter slot i (hd_train_param_slot order)
 *   hd_train_stage_wait    make `stream` (a hipStream_t) wait for stage k of the LAST hd_train_loss_backward: hipStreamWaitEvent, no host wait */
int hd_train_stage_count(const hd_trainer* t);
int hd_train_slot_stage(const hd_trainer* t, int slot, int* stage);
int hd_train_stage_wait(hd_trainer* t, int stage, void* stream);

/* torch.optim.Adam without weight decay / amsgrad over one flat array, one launch: g = grads * grad_scale (1/world after a
 * summing all-reduce); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps).
 * step counts from 1.  No context needed. */
int hd_adam_step(float* params, const float* grads, float* m, float* v, long long n, float lr, float b1, float b2, float eps, int step,
                 float grad_scale, void* stream);

/* ---- measurement ---------------------------------------------------------------------------- */

/* Per-launch HIP-event timing of the convolution kernels on their launch stream (process-wide;
 * not for use under stream capture).  hd_profile_enable(1) clears and starts recording,
 * hd_profile_read synchronises on the recorded events and fills one row per kernel instantiation
 * (named exactly as rocprofv3 --kernel-trace prints it, e.g. "conv_igemm_bf16x3_kernel<4, 1, 2, 2, 32, 6, 0, 9>")
 * with the number of launches, their summed duration and their summed ALGORITHMIC flops / bytes (each
 * conv reads its input once, writes its output once, reads its weights once); it returns the number of
 * rows filled (<= max_rows, <= HD_PROFILE_MAX_ROWS) or a negative error. bench.py derives
 * roofline.achieved from it. */
typedef struct {
    const char* kernel;
    long long launches;
    double total_ms;
    double flops;
    double bytes;
} hd_profile_row;
#define HD_PROFILE_MAX_ROWS 96
int hd_profile_enable(int enable);
int hd_profile_read(hd_profile_row* rows, int max_rows);

#ifdef __cplusplus
}
#endif
#endif /* HICDIFF_HIP_H */
