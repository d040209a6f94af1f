// Shared declarations of the gfx950 HiCDiff engine (internal; the public surface is include/hicdiff_hip.h).
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

// Activations are NHWC fp32: [B][H][W][C].  The network's own input/output tiles have C == 1, so
// the boundary's (B,1,S,S) tensors are this layout already.
struct Act {
    float* p = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    size_t numel() const { return (size_t)B * H * W * C; }
    size_t pixels() const { return (size_t)B * H * W; }
};

// Packed convolution weight: [KH*KW][Cin][CoutPad] fp32 (CoutPad = Cout rounded up to 64, zero
// filled), bias [Cout].  UNet 3x3 convs are weight-standardised at pack time.
struct ConvW {
    float* w = nullptr;
    unsigned short* wsplit = nullptr;   // split-bf16 image [taps][Cin/16][CoutPad][16 hi | 16 lo] (fast path; always 16-channel k-steps)
    unsigned short* wsplit16 = nullptr; // 3x3 only: the same layout with fp16 hi | lo of the exact weight (two-product arithmetic, conv_bf16x3_kernel.h AR = 2)
    unsigned short* wino = nullptr;     // 3x3 only: Winograd F(2x2,3x3) filter transform, split bf16, MFMA-fragment order (conv_winograd.hip)
    float* bias = nullptr;
    int KH = 1, KW = 1, Cin = 0, Cout = 0, CoutPad = 0, ck = 16;   // ck: activation slice of the bf16x3 kernel (32 where the channel counts allow)
    int f16_range_ok = 1;             // the fp16 hi | lo image is faithful: max |w| inside [2^-8, 2^15] (engine.hip Loader::finish_f16_range); else the layer keeps three products
};

enum { HD_PREC_F32 = 0, HD_PREC_BF16X3 = 1 };

// Per-step scalars of a fused sampler step, resident in device memory when the step is replayed from a
// hipGraph: f[0] = time value; ddpm: f[1..5] = recip, recipm1, coef1, coef2, sigma; ddrm: f[1..8] =
// sqrt_at, sqrt_1m_at, sqrt_at_next, sigma_next, sigma_0, etaA, etaB, etaC; ddpm also f[6] = weight of eps itself (DDIM steps).
struct StepParams { float f[12]; uint32_t step; uint32_t pad; uint64_t seed; uint64_t tile_off; };

enum InMode { IN_NONE = 0, IN_AFFINE_SILU = 1, IN_LAYERNORM = 2, IN_SOFTMAX32 = 4 };   // 4: softmax over every 32-channel group (LinearAttention q, src/hicdiff.py:217); bf16x3 kernel only
enum EpFlags {
    EP_FILM_SILU = 1,      // v = silu(v * (scale[b][n] + 1) + shift[b][n])       (hicedrn block, first conv)
    EP_ADD_SILU = 2,       // v = silu(v + shift[b][n])                            (hicedrn SR3 block)
    EP_RES = 4,            // v = alpha * v + res[pix][n]
    EP_RES_AFFINE_SILU = 8, // v = v + silu(res[pix][n] * resA[b][n] + resB[b][n]) (UNet block tail through res_conv)
    EP_FILM_SILU_BWD = 64, // training, data gradient through a = silu(u (scale + 1) + shift): v is dL/da / alpha; with u = res[pix][n]:
                           // dv = alpha v silu'(u (scale + 1) + shift), out = dv (scale + 1) (epScale null: scale = 0); with gn_part the
                           // per-channel sums over the tile's pixels (sum dv u, sum dv) go where the GroupNorm sums would (d scale, d shift)
    EP_LN_STATS = 32,      // also write the channel-LayerNorm statistics (mean, rstd) of the final row to ln_stats_out[pix]; needs Cout == tile width
    EP_LN_RES = 16         // v = LayerNorm_channels(v) * ep_ln_g[n] + res[pix][n]   (LinearAttention to_out tail, src/hicdiff.py:207-210,64-70); needs Cout == tile width
};

// GroupNorm finalize folded into the producer (conv epilogue / split-K reduce) where one workgroup sees all pixels of a sample for its
// channels: what gn_finalize_kernel would compute from the partial sums -- the per-(sample, channel) affine A, Bv [, E] -- written directly.
// A == nullptr: not requested.
struct GnFinArgs {
    const float* gamma = nullptr; const float* beta = nullptr; const float* film = nullptr;
    int film_bs = 0, film_off = 0, film_mode = 0, groups = 8;
    float* A = nullptr; float* Bv = nullptr; float* E = nullptr;
};

struct ConvArgs {
    // input (optionally the channel concat of two tensors, never materialised)
    const float* in0 = nullptr; const float* in1 = nullptr;
    int C0 = 0, C1 = 0;
    int B = 0, H = 0, W = 0;      // output spatial size
    int IH = 0, IW = 0;           // size of the stored input tensor(s)
    int stride = 1, pad = 0, upsample = 0;
    ConvW cw;
    // transform applied while staging the input tile
    int in_mode = IN_NONE;
    const float* inA = nullptr; const float* inB = nullptr; const float* inE = nullptr;  // [Bs][Cin]
    int in_bstride = 0;           // elements between samples in inA/inB/inE (0: broadcast one row)
    const float* ln_stats = nullptr;  // [pixels][2] mean, rstd
    const float* ln_g = nullptr;      // [Cin]
    // epilogue
    int ep = 0;
    const float* epScale = nullptr; const float* epShift = nullptr; int ep_bstride = 0;
    float alpha = 1.f;
    const float* res = nullptr; const float* resA = nullptr; const float* resB = nullptr; int res_bstride = 0;
    const float* ep_ln_g = nullptr;   // EP_LN_RES: LayerNorm gain [Cout]
    float* ln_stats_out = nullptr;    // EP_LN_STATS: [pixels][2]
    size_t w_bstride = 0;             // bf16x3 only: bytes between per-sample images of cw.wsplit (0: one shared weight); forces one sample per tile
    float* out = nullptr;
    float* pre_out = nullptr;         // EP_FILM_SILU / EP_ADD_SILU: also store the value BEFORE the FiLM + SiLU (bias added) here -- the training step keeps both
    // optional per-channel partial sums of the (pre-activation) output for GroupNorm:
    // gn_part[b][slot][Cout][2]; slots per sample = gn_slots (filled by the launcher)
    float* gn_part = nullptr;
    GnFinArgs gn_fin;                 // with gn_part: also finalize in the kernel when it can (launch_conv reports whether it did)
    // split-K workspace (conv_splitk(a) * B*H*W*Cout floats) for the convolutions the planner splits; null: never split
    float* splitk_ws = nullptr;
    int f16w2 = 0;                    // with precision == HD_PREC_BF16X3: 3x3 layers that carry cw.wsplit16 take two fp16 products per multiply
    int plain_bf16 = 0;               // with precision == HD_PREC_BF16X3: drop the two correction products where a plain-bf16 variant exists (training option)
    // N-tile rule override (0: the default): feature maps of at most this many pixels take 64-wide N tiles.  The training steps set 64
    // (64 tiles per step: the 8x8 maps' 128-wide tiles fill a quarter of the chip; 28.1 -> 26.0 ms per UNet step); a rule by map size only.
    int narrow_max_hw = 0;
    int precision = HD_PREC_F32;
};

// A size-prefixed struct of the ABI (include/hicdiff_hip.h) -> a zero-filled copy of this build's revision.  Exactly `struct_bytes` bytes of
// the caller's memory are read; sizes this build does not know -- a struct from a binding without the prefix shows up here as the bit
// pattern of its first float -- are refused, never read past.
template <typename T>
static inline bool hd_read_prefixed(const T* src, size_t oldest_bytes, T* dst) {
    uint32_t n;
    memcpy(&n, src, sizeof(n));
    if (n < oldest_bytes || n > sizeof(T) || n % 4) return false;
    memset(dst, 0, sizeof(T));
    memcpy(dst, src, n);
    return true;
}

// hipFuncAttributeMaxDynamicSharedMemorySize once per (device, kernel): the attribute is per device, a second context on another card of
// the same process needs it too.  false: the call failed.
bool hd_raise_dynamic_lds(const void* kernel, int bytes);

#define HD_CHECK_HIP(expr)                                                                 \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            hd_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));               \
            return -3;                                                                     \
        }                                                                                  \
    } while (0)

void hd_set_error(const std::string& msg);

void hd_prof_enable(bool on);
bool hd_prof_is_on();
void hd_prof_begin(const char* name, double flops, double bytes, hipStream_t st);   // + conv_prof_end(st) after the launch
void conv_prof_end(hipStream_t st);
int hd_prof_collect(const char** names, double* ms, double* flops, double* bytes, long long* launches, int max_rows);   // returns the row count

// ---- launchers (each only enqueues on `st`) ---------------------------------------------------
int launch_conv(const ConvArgs& a, hipStream_t st, int* gn_slots_out = nullptr);
bool conv_gn_direct(const ConvArgs& a);   // with a.gn_fin set: launch_conv writes the GroupNorm affine itself (no gn_finalize launch needed)
int conv_splitk(const ConvArgs& a);    // K splits the planner wants for this convolution (1: none); a.precision must be set
bool conv_film_bwd_ok(const ConvArgs& a);   // can this launch run the EP_FILM_SILU_BWD epilogue (8-wave 3x3 tile only)?
int conv_gn_slots(const ConvArgs& a);  // slots per sample the fused GN partials would use (0: not fusable)

int launch_pack_conv(const float* src_oihw, float* dst, int Cout, int Cin, int KH, int KW, int CoutPad,
                     int standardize, int unshuffle, hipStream_t st);
int launch_split_conv(const float* packed, unsigned short* dst, int taps, int Cin, int CoutPad, int CK, hipStream_t st, int f16 = 0, unsigned* absmax = nullptr);   // absmax (f16 only): atomicMax of the bits of |w| over the filter
size_t conv_winograd_weight_bytes(int Cin, int CoutPad);
int launch_pack_winograd(const float* packed, unsigned short* dst, int Cin, int CoutPad, hipStream_t st);   // packed: fp32 [9][Cin][CoutPad]
bool conv_uses_winograd(const ConvArgs& a);   // a.precision must be set
void conv_set_winograd(int mode);             // -1: HICDIFF_WINOGRAD decides (default off), 0: off, 1: on
bool conv_winograd_enabled();                 // the switch as it stands now (hd_load_weights packs the Winograd filter images only while it is on)
int launch_transpose(const float* src, float* dst, int rows, int cols, int dst_ld, int dst_col0, hipStream_t st);

int launch_conv_small_cin(const float* x, const float* cond, const float* w, const float* bias, float* out,
                          int B, int S, int KS, int Cin, int Cout, hipStream_t st, bool lanes_ok = false);   // lanes_ok: see small_kernels.hip
int launch_rowdot(const float* x, const float* w, const float* bias, float* out, size_t P, int C, hipStream_t st);

int launch_time_mlp(const void* t, int t_kind, float tval, const StepParams* sp, int sr3, int Bt, int dim, int time_dim, const float* w1t,
                    const float* b1, const float* w3t, const float* b3, float* temb, float* temb_act, hipStream_t st);
int launch_film(const float* act, int Bt, int K, const float* wt, const float* bias, int N, float* out, hipStream_t st);
// both of the above in one launch for the sampling loops' single time row (time_dim % 256 == 0 up to 1024, dim <= 256, dim % 16 == 0, N % 4 == 0)
int launch_time_film(float tval, const StepParams* sp, int sr3, int dim, int time_dim, const float* w1t, const float* b1, const float* w3t,
                     const float* b3, const float* wt, const float* bias, int N, float* out, hipStream_t st);

int launch_gn_partial(const float* x, int B, int HW, int C, float* part, int* slots_out, hipStream_t st);
int launch_gn_finalize(const float* part, int slots, int B, int HW, int C, int groups, const float* gamma,
                       const float* beta, const float* film, int film_bstride, int film_off, int film_mode,
                       float* A, float* Bv, float* E, hipStream_t st, float* stats_out = nullptr);   // stats_out: [B][groups][2] (mean, rstd), training only
int launch_affine_silu_add(const float* h, const float* A, const float* Bv, const float* res, float* out, int B, int HW, int C,
                           hipStream_t st, float* stats = nullptr);   // stats: see small_kernels.hip; returns 1 when it wrote them
int launch_ln_stats(const float* x, size_t P, int C, float* stats, hipStream_t st);
int launch_ln_residual(const float* y, const float* g, const float* res, float* out, size_t P, int C, hipStream_t st);
size_t linattn_scratch_floats(int B, int HW, int heads);
int launch_linattn_context(const float* qkv, int B, int HW, int heads, float* scratch, float* ctx, hipStream_t st);
int launch_linattn_apply(const float* qkv, int qstride, const float* ctx, int B, int HW, int heads, float* out, hipStream_t st);
int launch_linattn_combine(const float* pmax, const float* psum, const float* pctx, int B, int heads, int nsplit, int HW, float* ctx,
                           hipStream_t st);
int launch_pack_kv(const float* wqkv, const float* g, int C, unsigned short* dst, hipStream_t st);
int linattn_kv_nsplit(int HW);
int linattn_kv_nparts(int HW, int C);   // partials per (sample, head) written by launch_linattn_kv_fused (<= linattn_kv_nsplit: the 64-channel kernel merges its chunks)
int launch_linattn_kv_fused(const float* x, const unsigned short* wkv, int B, int HW, int C, float* pmax, float* psum, float* pctx,
                            hipStream_t st);
int launch_pack_q(const float* wqkv, const float* g, int C, unsigned short* dst, hipStream_t st);
int launch_linattn_q_fused(const float* x, const float* stats, const unsigned short* wq, const unsigned short* wfold, const float* bias,
                           const float* gout, float* out, int B, int HW, int C, hipStream_t st);
int launch_linattn_fold_out(const float* wout_packed, const float* ctx, int B, int CoutPad, unsigned short* dst, hipStream_t st, int permute = 0);
int launch_split_pieces(const float* mat, int n, const int* origins, int ntiles, int piece, float* tiles, hipStream_t st);
int launch_stitch_pieces(const float* tiles, const int* tile_of, int nb, int piece, int step, float* mat, int n, hipStream_t st);
int launch_tile_metrics(const float* pred, const float* target, int B, int S, int rescale, double* partial, double* sums, float* ssim_each,
                        hipStream_t st);
// training components (train_norms.hip)
size_t gn_bwd_scratch_floats(int B, int HW, int C);
// g -> gout / dy -> dout: the output may be the input buffer (in place) or another one
int launch_gn_silu_bwd(const float* x, const float* g, float* gout, const float* A, const float* Bv, const float* stats, const float* gamma, const float* beta,
                       const float* film, int film_bs, int film_off, int film_mode, int B, int HW, int C, int G, float* scratch, float* dgamma, float* dbeta,
                       float* dfilm, int accumulate, hipStream_t st, int dfilm_bs = 0);
size_t ln_bwd_scratch_floats(size_t P, int C);
int launch_ln_bwd(const float* x, const float* dy, float* dout, const float* gain, size_t P, int C, float* scratch, float* dgain, int accumulate, hipStream_t st,
                  const float* add = nullptr);   // add (may be dout): dout = add + dx
int launch_ws_bwd(const float* w, const float* dwhat, int Cout, int n, float* dw, hipStream_t st);
int launch_ws_fwd(const float* w, int Cout, int n, float* out, hipStream_t st);
int launch_attn_full_bwd(const float* qkv, const float* dout, int B, int n, int heads, float* dqkv, hipStream_t st);   // train_attn.hip
int launch_first_conv_wgrad(const float* g, const float* in0, const float* in1, int J, int B, int S, int C, int KS, float* scratch, float* dW, hipStream_t st);
int launch_rowdot_bwd(const float* x, const float* dout, const float* w, size_t P, int C, float* scratch, float* dx, float* dw, hipStream_t st);
int launch_sum_pool2(const float* g, int B, int H, int W, int C, float* dx, hipStream_t st);        // train.hip: Upsample backward (g is [B][2H][2W][C])
int launch_pixel_shuffle(const float* g, int B, int H, int W, int C, float* dx, hipStream_t st);    // Downsample backward (g is [B][H][W][4C])
size_t linattn_bwd_scratch_floats(int B, int n, int heads);
int launch_linattn_bwd(const float* qkv, const float* dout, int B, int n, int heads, float* scratch, float* dqkv, hipStream_t st);
int launch_attn_full(const float* qkv, int B, int HW, int heads, float* out, hipStream_t st);

int launch_ddpm_update(float* x, const float* eps, const float* noise, float c_recip, float c_recipm1, float coef1,
                       float coef2, float sigma, float* x0_out, int B, int S, uint64_t seed, uint64_t tile_off,
                       uint32_t step, const StepParams* sp, hipStream_t st, float coef_eps = 0.f, uint32_t tile_add = 0);   // tile_add: added to sp's tile offset (chained steps)
int launch_ddrm_update(float* x, const float* eps, const float* y, const float* z, float sqrt_at, float sqrt_1m_at,
                       float sqrt_at_next, float sigma_next, float sigma_0, float etaA, float etaB, float etaC,
                       float* x0_out, int B, int S, uint64_t seed, uint64_t tile_off, uint32_t step, const StepParams* sp,
                       hipStream_t st, uint32_t tile_add = 0);
int launch_set_step_params(StepParams* dst, const StepParams& v, hipStream_t st);
int launch_spin_us(int us, hipStream_t st);
int launch_q_sample(const float* x0, const float* noise, const float* a, const float* s, float* out, int B, int S,
                    hipStream_t st);
int launch_loss(const float* pred, const float* target, int l2, float* out, int B, int S, hipStream_t st);
int launch_randn(float* out, int B, int S, uint64_t seed, uint64_t tile_off, uint32_t step, hipStream_t st, uint32_t nstream = 0);
