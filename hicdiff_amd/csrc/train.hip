// Native training step of the hicedrn eps-network: p_losses forward with saved
// activations, hand-written backward, fused Adam.  Reference: train.py:109-190 (loss = diffusion(x); loss.backward();
// optim.Adam(lr=2e-5).step()), src/hicdiff.py:711-755, src/hicdiff_condition.py:715-750, src/model/hicedrn_Diff.py:169-289.
//
// Convolutions: the forward and the data gradient run on the implicit-GEMM kernel of conv_bf16x3_kernel.h (the data
// gradient is the same convolution with the weight flipped and transposed).  The weight gradient is a split-bf16 x3 NT
// GEMM over pixels: both operands are first rewritten channel-major as bf16 hi / lo images over a PADDED pixel axis
// (one zero row per sample, eight zero columns per row), so that the nine taps become nine constant shifts of one flat
// axis: the row shift is applied to the activation pointer, the column shift is baked into three copies of the
// gradient operand.  Everything is deterministic (split-K partials are reduced in a fixed order, no atomics).
#include "hd_common.h"
#include "../../include/hicdiff_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <deque>
#include <map>
#include <string>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string(what) + ": " + hipGetErrorString(e)); return -3; }
    return 0;
}

// ---- weight-gradient operands ---------------------------------------------------------------------------------------
// in: NHWC fp32 [B][H][W][C], C <= 256, W <= 64, W % 8 == 0.  out: bf16 hi / lo images [C][ld]; pixel (b,y,x) of channel c
// lands at k = guard + (b*(H+1) + y + 1) * P + 8 + x.  Pad positions are never written: the buffers are zeroed once.
// mode 1 applies the block's FiLM + SiLU on the way in (the activation between the two uses of the block's convolution is
// recomputed instead of stored).  colpart (optional): per-row channel sums for the bias gradient.
// One workgroup per image row: float4 loads of four channels per thread on the way in, LDS transpose, then 16-byte stores of
// 8 pixels per lane on the way out (8 lanes cover a channel's 64-pixel row segment).
template <bool PLAIN>      // PLAIN: hi image only (the optional bf16 arithmetic)
__global__ __launch_bounds__(256) void wg_prep_kernel(const float* __restrict__ in, int C, int H, int W, int P, size_t ld, size_t guard,
                                                      unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, int mode,
                                                      const float* __restrict__ film, int film_bs, float* __restrict__ colpart, int row0,
                                                      const float* __restrict__ affB, int src_mode, const float* __restrict__ affE) {
    // src_mode 0: `in` is [B][H][W][C].  1: nearest x2 upsample on the way in (Upsample, src/hicdiff.py:72-76): `in` is [B][H/2][W/2][C].
    // 2: pixel-unshuffle on the way in (Downsample, src/hicdiff.py:78-82, 'b c (h p1) (w p2) -> b (c p1 p2) h w'): `in` is [B][2H][2W][C/4]
    //    and logical channel cc = c*4 + p1*2 + p2 reads (2y + p1, 2x + p2, c).
    // affE (mode 2 only, may be null): + affE[b*film_bs + c] AFTER the SiLU -- the SR3 blocks' additive noise embedding (src/hicdiff_sr3.py:246-251).
    // mode 0: raw; mode 1: silu(v * (film[b][c] + 1) + film[b][film_bs - C + c]) (film_bs == C: no scale) -- the hicedrn block;
    // mode 2: silu(v * film[b*film_bs + c] + affB[b*film_bs + c]) -- a GroupNorm'd, FiLM'd activation given as a per-(sample, channel) affine.
    // grid.y walks the channels in chunks of 256; row0 = first image row this tensor's channels go to (channel-concatenated inputs).
    constexpr int PITCH = 68;                                  // ushorts per channel row: 136 B, 8-byte aligned, 2-way conflicts at most
    __shared__ __attribute__((aligned(16))) unsigned short sh_hi[256 * PITCH], sh_lo[256 * PITCH];
    __shared__ float csum[1024];                                // [pixel groups][channels of this chunk]
    const int b = blockIdx.x / H, y = blockIdx.x % H, tid = threadIdx.x, cbase = blockIdx.y * 256;
    const int W8 = (W + 7) & ~7;
    // thread = (channel quad, pixel group): float4 loads; a chunk of cc channels has cc/4 quads and 256 / (cc/4) pixel groups, so narrow
    // layers (64, 128 channels -- the full-resolution levels of the UNet) keep all 256 threads loading
    const int cc = min(256, C - cbase), nq = cc >> 2, npg = 256 / nq;       // cc is a multiple of 4 and, for cc < 256, a power-of-two multiple of 64 in practice
    const int cq = tid % nq, pgp = tid / nq, c0 = cq * 4, cg = cbase + c0;
    if (pgp < npg) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), shf = make_float4(0.f, 0.f, 0.f, 0.f), sum = make_float4(0.f, 0.f, 0.f, 0.f), add = shf;
        if (mode == 1) {                                       // film row: [scale | shift] (film_bs == 2C) or [shift] alone (SR3, film_bs == C)
            if (film_bs == 2 * C) {
                sc = *reinterpret_cast<const float4*>(film + (size_t)b * film_bs + cg);
                sc.x += 1.f; sc.y += 1.f; sc.z += 1.f; sc.w += 1.f;
            }
            shf = *reinterpret_cast<const float4*>(film + (size_t)b * film_bs + (film_bs - C) + cg);
        } else if (mode == 2) {
            sc = *reinterpret_cast<const float4*>(film + (size_t)b * film_bs + cg);
            shf = *reinterpret_cast<const float4*>(affB + (size_t)b * film_bs + cg);
            if (affE) add = *reinterpret_cast<const float4*>(affE + (size_t)b * film_bs + cg);
        }
        const float* src = in + ((size_t)(b * H + y) * W) * C + cg;
        if (src_mode == 1) src = in + ((size_t)(b * (H / 2) + (y >> 1)) * (W / 2)) * C + cg;
        const int Cs = C / 4, cs = cg >> 2;                    // unshuffle: source channels, this quad's source channel
        for (int x = pgp; x < W8; x += npg) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);       // columns W .. W8-1 of the last 8-pixel group stay zero (they are padding)
            if (x < W) {
                if (src_mode == 2) {
                    const float* s0 = in + (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * Cs + cs;
                    v = make_float4(s0[0], s0[Cs], s0[(size_t)2 * W * Cs], s0[(size_t)2 * W * Cs + Cs]);   // (p1,p2) = (0,0), (0,1), (1,0), (1,1)
                } else {
                    v = *reinterpret_cast<const float4*>(src + (size_t)(src_mode == 1 ? x >> 1 : x) * C);
                }
                if (mode) { v.x = silu_f(v.x * sc.x + shf.x) + add.x; v.y = silu_f(v.y * sc.y + shf.y) + add.y; v.z = silu_f(v.z * sc.z + shf.z) + add.z; v.w = silu_f(v.w * sc.w + shf.w) + add.w; }
            }
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const __bf16 hh = (__bf16)e[j];
                const __bf16 ll = (__bf16)(e[j] - (float)hh);
                sh_hi[(c0 + j) * PITCH + x] = __builtin_bit_cast(unsigned short, hh);
                if constexpr (!PLAIN) sh_lo[(c0 + j) * PITCH + x] = __builtin_bit_cast(unsigned short, ll);
            }
        }
        csum[pgp * cc + c0] = sum.x; csum[pgp * cc + c0 + 1] = sum.y; csum[pgp * cc + c0 + 2] = sum.z; csum[pgp * cc + c0 + 3] = sum.w;
    }
    __syncthreads();
    if (colpart && tid < cc) {
        float t = 0.f;
        for (int g = 0; g < npg; ++g) t += csum[g * cc + tid];
        colpart[(size_t)blockIdx.x * C + cbase + tid] = t;
    }
    const int xg = tid & 7;
    const size_t kbase = guard + ((size_t)b * (H + 1) + y + 1) * P + 8 + xg * 8;
    if (xg * 8 < W8) {
        for (int c = tid >> 3; c < 256 && cbase + c < C; c += 32) {
            const uint2* ph = reinterpret_cast<const uint2*>(sh_hi + c * PITCH + xg * 8);
            const uint2* pl = reinterpret_cast<const uint2*>(sh_lo + c * PITCH + xg * 8);
            const uint2 h0 = ph[0], h1 = ph[1];
            const size_t o = (size_t)(row0 + cbase + c) * ld + kbase;
            *reinterpret_cast<uint4*>(hi + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            if constexpr (!PLAIN) {
                const uint2 l0 = pl[0], l1 = pl[1];
                *reinterpret_cast<uint4*>(lo + o) = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        }
    }
}

// Weight-gradient GEMM, split-bf16 x3: partial[(split*3 + dyi)][ci][kx*F + co] = sum over this split's k of
// A[ci][k + (dyi-1)*P] * G[co][k - (kx-1)].  The three column taps are built in registers from one LDS image of G: a 16-byte operand (8 bf16 along k) shifted by one element is four v_alignbit over the aligned
// operand and one neighbouring dword.  Workgroup tile: 128 input channels x 64 output channels x 3 column taps; 4 waves as
// 2 (rows) x 2 (columns), each 64 x 32 x 3 taps = 6 accumulator tiles.  Per K slice of 64 the workgroup loads 32 KB of A and
// 20 KB of G (with an 8-element halo either side) for 72 MFMAs per wave.  The next slice is prefetched into NAMED registers under
// the MFMAs (an indexed register array went to scratch with a wait after every load: 4.6x slower), unconditionally (the last
// iteration re-reads its own slice).  Workgroups that stream the same k range share an XCD (blockIdx round-robins over the 8
// XCDs), so the 8 tiles of a split re-read its operand slices from that XCD's L2.
__device__ __forceinline__ uint4 shift_prev(const uint4& c, unsigned prevw) {          // elements k-1 .. k+6
    return make_uint4(__builtin_amdgcn_alignbit(c.x, prevw, 16), __builtin_amdgcn_alignbit(c.y, c.x, 16),
                      __builtin_amdgcn_alignbit(c.z, c.y, 16), __builtin_amdgcn_alignbit(c.w, c.z, 16));
}
__device__ __forceinline__ uint4 shift_next(const uint4& c, unsigned nextx) {          // elements k+1 .. k+8
    return make_uint4(__builtin_amdgcn_alignbit(c.y, c.x, 16), __builtin_amdgcn_alignbit(c.z, c.y, 16),
                      __builtin_amdgcn_alignbit(c.w, c.z, 16), __builtin_amdgcn_alignbit(nextx, c.w, 16));
}

template <bool PLAIN, bool ONE, int TMW = 2>      // PLAIN: hi x hi only -- no lo images are loaded, staged or multiplied.  ONE: 1x1 filter, centre tap only.
// TMW: 32-row blocks per wave: 2 = 128 input channels per workgroup, 1 = 64 (layers with 64 input channels: no padded half)
__global__ __launch_bounds__(256, 2) void wgrad_gemm_kernel(const unsigned short* __restrict__ Ahi, const unsigned short* __restrict__ Alo,
                                                            const unsigned short* __restrict__ Ghi, const unsigned short* __restrict__ Glo,
                                                            size_t ld, size_t guard, int P, int slices_per_split, int total_slices, int nsplit, int Mtiles,
                                                            int Ntiles, int M, int F, float* __restrict__ partial, int xcd_group) {
    constexpr int BM = 64 * TMW, BN = 64, KS = 64, PA = KS * 2 + 16, PG = (KS + 16) * 2 + 16;      // 144, 176 bytes
    __shared__ __attribute__((aligned(16))) char sA[2][BM * PA];
    __shared__ __attribute__((aligned(16))) char sG[2][BN * PG];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = w >> 1, wn = w & 1;
    const int per = Mtiles * Ntiles;
    int split, inner;
    if ((nsplit & 7) == 0 && xcd_group) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; split = xcd + 8 * (j / per); inner = j % per; }
    else { split = blockIdx.x / per; inner = blockIdx.x % per; }
    const int nt = inner % Ntiles, mt = inner / Ntiles;
    // this split's k range: slices_per_split slices of 64, the last split takes what is left (at least one: the host sizes nsplit so)
    const size_t k0 = guard + (size_t)split * slices_per_split * KS;
    const int nslices = min(slices_per_split, total_slices - split * slices_per_split);
    constexpr int NDX = ONE ? 1 : 3;
    const int N = NDX * F;
    // A: rows lrow + 32 j (j < 4), chunk lch.  G: 64 rows x 10 chunks x {hi, lo} = 1280 chunks, five per thread.
    const int lrow = tid >> 3, lch = tid & 7;
    const size_t aoff = (size_t)lrow * ld + lch * 8;
    const int aloff = lrow * PA + lch * 16;
#define HD_GDEF(j)                                                                                                   \
    const int q##j = tid + 256 * j, wh##j = q##j >= 640 ? 1 : 0, rem##j = q##j - 640 * wh##j, row##j = rem##j / 10,  \
              ch##j = rem##j - 10 * row##j;                                                                          \
    const unsigned short* pg##j = (wh##j ? Glo : Ghi) + (size_t)(nt * BN + row##j) * ld + (k0 - 8) + ch##j * 8;      \
    char* const lg##j = sG[wh##j] + row##j * PG + ch##j * 16;
    HD_GDEF(0) HD_GDEF(1) HD_GDEF(2) HD_GDEF(3) HD_GDEF(4)
#undef HD_GDEF
    // the three row taps are three workgroups (blockIdx.y): same XCD when gridDim.x is a multiple of 8, so they share the operand slices in L2
    const int dyi = ONE ? 1 : (int)blockIdx.y;
    {
    const size_t ka = (size_t)((long)k0 + (long)(dyi - 1) * P);
    const unsigned short* pAh = Ahi + (size_t)mt * BM * ld + ka + aoff;
    const unsigned short* pAl = Alo + (size_t)mt * BM * ld + ka + aoff;
    uint4 rAh0, rAh1, rAh2, rAh3, rAl0, rAl1, rAl2, rAl3, rG0, rG1, rG2, rG3, rG4;
#define HD_LD(p, j, K) (*reinterpret_cast<const uint4*>((p) + (size_t)(j) * 32 * ld + (K)))
#define HD_LG(j, K) (*reinterpret_cast<const uint4*>(pg##j + (K)))
#define HD_GLOAD(K)                                                                                  \
    rAh0 = HD_LD(pAh, 0, K); rAh1 = HD_LD(pAh, 1, K);                                                       \
    if constexpr (TMW == 2) { rAh2 = HD_LD(pAh, 2, K); rAh3 = HD_LD(pAh, 3, K); }                           \
    if constexpr (!PLAIN) {                                                                                 \
        rAl0 = HD_LD(pAl, 0, K); rAl1 = HD_LD(pAl, 1, K);                                                   \
        if constexpr (TMW == 2) { rAl2 = HD_LD(pAl, 2, K); rAl3 = HD_LD(pAl, 3, K); }                       \
    }                                                                                                       \
    rG0 = HD_LG(0, K); rG1 = HD_LG(1, K); rG2 = HD_LG(2, K);                                                \
    if constexpr (!PLAIN) { rG3 = HD_LG(3, K); rG4 = HD_LG(4, K); }
#define HD_STA(m, j, r) *reinterpret_cast<uint4*>(sA[m] + aloff + (j) * 32 * PA) = r
#define HD_STG(j, r) *reinterpret_cast<uint4*>(lg##j) = r
    f32x16 acc[TMW][NDX];
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int b = 0; b < NDX; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    HD_GLOAD(0)
    for (int s = 0; s < nslices; ++s) {
        HD_STA(0, 0, rAh0); HD_STA(0, 1, rAh1);
        if constexpr (TMW == 2) { HD_STA(0, 2, rAh2); HD_STA(0, 3, rAh3); }
        if constexpr (!PLAIN) {
            HD_STA(1, 0, rAl0); HD_STA(1, 1, rAl1);
            if constexpr (TMW == 2) { HD_STA(1, 2, rAl2); HD_STA(1, 3, rAl3); }
        }
        HD_STG(0, rG0); HD_STG(1, rG1); HD_STG(2, rG2);          // chunks 0..767: all of hi (and, harmlessly, the first lo chunks)
        if constexpr (!PLAIN) { HD_STG(3, rG3); HD_STG(4, rG4); }
        __syncthreads();
        { const size_t kn = (size_t)min(s + 1, nslices - 1) * KS; HD_GLOAD(kn) }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 ah[TMW], al[TMW];
#pragma unroll
            for (int t = 0; t < TMW; ++t) {
                const int ro = (wm * 32 * TMW + t * 32 + l31) * PA + ks * 32 + half * 16;
                ah[t] = *reinterpret_cast<const bf16x8*>(sA[0] + ro);
                if constexpr (!PLAIN) al[t] = *reinterpret_cast<const bf16x8*>(sA[1] + ro);
            }
            const int go = (wn * 32 + l31) * PG + ks * 32 + half * 16 + 16;                 // element k sits at byte (k - k0 + 8) * 2
            const uint4 ch = *reinterpret_cast<const uint4*>(sG[0] + go);
            const unsigned ph = *reinterpret_cast<const unsigned*>(sG[0] + go - 4), nh = *reinterpret_cast<const unsigned*>(sG[0] + go + 16);
            bf16x8 bh[3], bl[3];
            // kx = 0 (dx = -1): G[k + 1 ...];  kx = 1: G[k ...];  kx = 2 (dx = +1): G[k - 1 ...]
            if constexpr (ONE) {
                bh[0] = __builtin_bit_cast(bf16x8, ch);
                if constexpr (!PLAIN) bl[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sG[1] + go));
            } else {
                bh[0] = __builtin_bit_cast(bf16x8, shift_next(ch, nh));
                bh[1] = __builtin_bit_cast(bf16x8, ch);
                bh[2] = __builtin_bit_cast(bf16x8, shift_prev(ch, ph));
                if constexpr (!PLAIN) {
                    const uint4 cl = *reinterpret_cast<const uint4*>(sG[1] + go);
                    const unsigned pl = *reinterpret_cast<const unsigned*>(sG[1] + go - 4), nl = *reinterpret_cast<const unsigned*>(sG[1] + go + 16);
                    bl[0] = __builtin_bit_cast(bf16x8, shift_next(cl, nl));
                    bl[1] = __builtin_bit_cast(bf16x8, cl);
                    bl[2] = __builtin_bit_cast(bf16x8, shift_prev(cl, pl));
                }
            }
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
                for (int dx = 0; dx < NDX; ++dx) {
                    if constexpr (!PLAIN) {
                        acc[tm][dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[dx], acc[tm][dx], 0, 0, 0);
                        acc[tm][dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[dx], acc[tm][dx], 0, 0, 0);
                    }
                    acc[tm][dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[dx], acc[tm][dx], 0, 0, 0);
                }
        }
        __syncthreads();
    }
#undef HD_GLOAD
#undef HD_LD
#undef HD_LG
#undef HD_STA
#undef HD_STG
    float* out = partial + ((size_t)(split * NDX + (ONE ? 0 : dyi)) * M) * N;
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
        for (int dx = 0; dx < NDX; ++dx)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * BM + wm * 32 * TMW + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int n = dx * F + nt * BN + wn * 32 + l31;
                out[(size_t)m * N + n] = acc[tm][dx][r];
            }
    }
}

// ---- weight gradient straight from the NHWC fp32 tensors (no operand rewrite) ---------------------------------------------------
// Same product and the same partial layout as wgrad_gemm_kernel, but the operands are read where the forward left them:
// activations X [B][H][W][C0 (+ C1)] (channel concat = two pointers) and output gradient G [B][H][W][F], fp32.  k runs over a padded
// pixel space of pitch W + 1 with H + 1 rows per sample (one zero column / row is all the isolation the 3 x 3 taps need), 64 positions
// per slice.  The loader (thread = 4 pixels x one channel quad of G and one or two of X) splits each float4 into bf16 hi / lo on the
// fly and writes [pixel][channel] images into LDS; the MFMA fragments -- 8 consecutive k per lane, i.e. a COLUMN of that image --
// come out of it with ds_read_b64_tr_b16 (tools/tr_read_probe.hip pins the lane mapping).  The column taps are whole-row offsets of
// the G image (pixel k + 1 - kx), the row tap dy is a shift of the X source row, zero outside the map.  The four pixel rows of a
// transposed read must fall on disjoint bank quarters: G rows are 128 + 64 bytes, X rows are swizzled in 64-byte chunks.
// Bias gradient: the workgroups of the first row tile (centre row tap) also sum their G slices per channel -> biaspart[split][F].
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct WgdArgs {
    const float* x0; const float* x1; int C0, C1;
    const float* g; int F;
    int B, H, W; unsigned magicW, magicH;                  // floor(n / (W+1)) = umulhi(n, magicW), same for H + 1
    int slices_per_split, total_slices, nsplit, Mtiles, Ntiles, M;
    float* partial; float* biaspart; int xcd_group;
    int src_mode;                                          // 0: X as [B][H][W][C]; 1: nearest x2 upsample of [B][H/2][W/2][C]; 2: pixel-unshuffle of [B][2H][2W][C/4]
    const float* affA; const float* affB; const float* affE; int aff_bs; float aff_addA;   // AFF kernels: X = silu(x * (affA[b][c] + aff_addA) + affB[b][c]) (+ affE[b][c])
};

// SiLU of the weight-gradient loaders: hardware reciprocal instead of the IEEE division (ten instructions per element in a phase where
// all eight waves of the workgroup do VALU work and the matrix pipe waits); 1 ulp from silu_f, far inside the bf16 x3 split's error
__device__ __forceinline__ float silu_rcp(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ uint2 split_quad(const float4& v, uint2& lo) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 h0 = {(__bf16)v.x, (__bf16)v.y}, h1 = {(__bf16)v.z, (__bf16)v.w};
    const bf2 l0 = {(__bf16)(v.x - (float)h0[0]), (__bf16)(v.y - (float)h0[1])}, l1 = {(__bf16)(v.z - (float)h1[0]), (__bf16)(v.w - (float)h1[1])};
    lo = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
    return make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
}
__device__ __forceinline__ bf16x8 tr_read8(const char* p, int second) {      // 8 consecutive pixels of this lane's channel: two 4-pixel blocks
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + second));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// AFF: 0 = X as stored; 1 = a normalised activation recomputed on the way in, silu(x A + B) per (sample, channel); 2 = the same + E after
// the SiLU (SR3's additive embedding).  A thread keeps ONE coefficient set per channel quad, that of its first pixel's sample, requested
// with the slice's prefetch; a pixel of another sample (a slice reaches into the next sample once per sample on large maps, in every
// slice on 8 x 8 maps) fetches its own at store time.  affA == null: A = 1; aff_addA is added to A (FiLM's scale + 1).
// SRC: source addressing of X (WgdArgs::src_mode) -- compiled in, the position arithmetic sits in the loop.
#ifdef HD_STAMPS
__device__ unsigned long long g_wgd_stamps[1024][5];      // per workgroup (wave 0): cycles in the store phase (incl. the wait for the prefetched loads), in barriers, in the MFMA phase, total
#endif
template <bool ONE, int TMW, int AFF = 0, int SRC = 0>
__global__ __launch_bounds__(256, 2) void wgrad_direct_kernel(WgdArgs a) {
    // bytes per pixel row: the G image is padded (its taps are row offsets), the X image is XOR-swizzled instead (never shifted): 64-byte
    // chunk index ^= pixel & 3 (256-byte rows) or ^= (pixel >> 1) & 1 (128-byte rows)
    constexpr int BM = 64 * TMW, PX = BM * 2, PG = 64 * 2 + 64;
    constexpr int XHI = 0, XLO = 64 * PX, GHI = 2 * 64 * PX, GLO = GHI + 66 * PG, LDSB = GLO + 66 * PG;
    __shared__ __attribute__((aligned(16))) char smem[LDSB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = w >> 1, wn = w & 1;
    const int per = a.Mtiles * a.Ntiles;
    int split, inner;
    if ((a.nsplit & 7) == 0 && a.xcd_group) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; split = xcd + 8 * (j / per); inner = j % per; }
    else { split = blockIdx.x / per; inner = blockIdx.x % per; }
    const int nt = inner % a.Ntiles, mt = inner / a.Ntiles;
    const int dyi = ONE ? 1 : (int)blockIdx.y, dy = dyi - 1;
    constexpr int NDX = ONE ? 1 : 3;
    const int Wp = a.W + 1, Hp = a.H + 1, Ktot = a.B * Hp * Wp, Cin = a.C0 + a.C1;
    const int s0 = split * a.slices_per_split;
    const int nslices = min(a.slices_per_split, a.total_slices - s0);
    const bool do_bias = a.biaspart != nullptr && mt == 0 && dyi == 1;

    // ---- loader roles: pixel group pg (pixels pg + 16 j of a slice), channel quad
    const int pg = tid >> 4, quad = tid & 15;
    const int gch = nt * 64 + quad * 4;                                          // F is a multiple of 64
    int xch[TMW]; const float* xsrc[TMW]; int xC[TMW]; bool xok[TMW];
#pragma unroll
    for (int t = 0; t < TMW; ++t) {
        const int c = mt * BM + t * 64 + quad * 4;
        xok[t] = c < Cin;
        if (c < a.C0) { xsrc[t] = a.x0 + c; xC[t] = a.C0; } else { xsrc[t] = a.x1 + (c - a.C0); xC[t] = a.C1; }
        if (!xok[t]) { xsrc[t] = a.x0; xC[t] = a.C0; }
        xch[t] = c;
    }
    float4 rX[4 * TMW], rG[4], rGh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cA[TMW], cB[TMW], cE[TMW];
    int s_first = 0; unsigned sdiff = 0;                                         // sample of pixel 0; (sample of pixel j) - s_first, 8 bits each
    unsigned vmask = 0;                                                          // bits 0-3: G item j valid; 4-7: X pixel j valid; 8: halo valid
    int sample = 0;
    auto locate = [&](int kp, int& gpix, bool& gv, int& xpix, bool& xv) {      // padded position -> source pixels
        const bool in = kp >= 0 && kp < Ktot;
        const unsigned k = in ? (unsigned)kp : 0u;
        const unsigned yq = __umulhi(k, a.magicW), xp = k - yq * Wp, b = __umulhi(yq, a.magicH), yp = yq - b * Hp;
        gv = in && (int)xp < a.W && (int)yp < a.H;
        const int ys = (int)yp + dy;
        xv = in && (int)xp < a.W && ys >= 0 && ys < a.H;
        gpix = ((int)b * a.H + (int)yp) * a.W + (int)xp;
        if constexpr (SRC == 1) xpix = ((int)b * (a.H >> 1) + (ys >> 1)) * (a.W >> 1) + ((int)xp >> 1);          // Upsample, src/hicdiff.py:72-76
        else if constexpr (SRC == 2) xpix = ((int)b * 2 * a.H + 2 * ys) * 2 * a.W + 2 * (int)xp;                  // top-left of the 2 x 2 source block
        else xpix = ((int)b * a.H + ys) * a.W + (int)xp;
        sample = in ? (int)b : 0;
    };
    struct Coef { float4 A, B, E; };
    auto coef = [&](int smp, int t) __attribute__((always_inline)) {            // the (sample, quad t) coefficients
        const size_t co = (size_t)smp * a.aff_bs + (xok[t] ? xch[t] : 0);
        Coef c;
        c.A = a.affA ? *reinterpret_cast<const float4*>(a.affA + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        c.A.x += a.aff_addA; c.A.y += a.aff_addA; c.A.z += a.aff_addA; c.A.w += a.aff_addA;
        c.B = *reinterpret_cast<const float4*>(a.affB + co);
        c.E = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (AFF == 2) c.E = *reinterpret_cast<const float4*>(a.affE + co);
        return c;
    };
    auto request = [&](int s) {
        const int k0 = (s0 + s) * 64;
        vmask = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int gp, xp; bool gv, xv;
            locate(k0 + pg + 16 * j, gp, gv, xp, xv);
            rG[j] = *reinterpret_cast<const float4*>(a.g + (size_t)(gv ? gp : 0) * a.F + gch);
            vmask |= (gv ? 1u : 0u) << j;
#pragma unroll
            for (int t = 0; t < TMW; ++t) {
                if constexpr (SRC == 2) {
                    // Downsample (src/hicdiff.py:78-82): logical channel c * 4 + p1 * 2 + p2 reads source pixel (2y + p1, 2x + p2), channel c:
                    // this quad is one source channel of the four pixels of the 2 x 2 block
                    const int Cs = a.C0 >> 2;
                    const float* sp = a.x0 + (size_t)(xv ? xp : 0) * Cs + (xok[t] ? xch[t] >> 2 : 0);
                    const size_t rowp = (size_t)2 * a.W * Cs;
                    rX[j * TMW + t] = make_float4(sp[0], sp[Cs], sp[rowp], sp[rowp + Cs]);
                } else {
                    rX[j * TMW + t] = *reinterpret_cast<const float4*>(xsrc[t] + (size_t)(xv ? xp : 0) * xC[t]);
                }
            }
            vmask |= (xv ? 1u : 0u) << (4 + j);
            if constexpr (AFF != 0) {
                if (j == 0) {
                    s_first = sample; sdiff = 0;
#pragma unroll
                    for (int t = 0; t < TMW; ++t) { const Coef c = coef(sample, t); cA[t] = c.A; cB[t] = c.B; cE[t] = c.E; }
                } else {
                    sdiff |= (unsigned)min(max(sample - s_first, 0), 255) << (8 * j);
                }
            }
        }
        if (!ONE && tid < 32) {                                                  // the G halo: positions k0 - 1 and k0 + 64
            int gp, xp; bool gv, xv;
            locate(tid < 16 ? k0 - 1 : k0 + 64, gp, gv, xp, xv);
            rGh = *reinterpret_cast<const float4*>(a.g + (size_t)(gv ? gp : 0) * a.F + gch);
            vmask |= (gv ? 1u : 0u) << 8;
        }
    };
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int xswz_w = (TMW == 2 ? (pg & 3) : ((pg >> 1) & 1)) << 6;             // pixel = pg + 16 j: its low bits are pg's
    auto store = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pl = pg + 16 * j;
            const float4 gq = (vmask >> j) & 1 ? rG[j] : zero4;
            if (do_bias) { bsum.x += gq.x; bsum.y += gq.y; bsum.z += gq.z; bsum.w += gq.w; }
            uint2 lo; const uint2 hi = split_quad(gq, lo);
            *reinterpret_cast<uint2*>(smem + GHI + (pl + 1) * PG + quad * 8) = hi;
            *reinterpret_cast<uint2*>(smem + GLO + (pl + 1) * PG + quad * 8) = lo;
#pragma unroll
            for (int t = 0; t < TMW; ++t) {
                float4 xq = rX[j * TMW + t];
                if constexpr (AFF != 0) {
                    float4 A = cA[t], Bc = cB[t], E = cE[t];
                    const int dsm = (int)((sdiff >> (8 * j)) & 255u);
                    if (dsm != 0) { const Coef c = coef(s_first + dsm, t); A = c.A; Bc = c.B; E = c.E; }   // a pixel of a later sample than the thread's first
                    xq = make_float4(silu_f(xq.x * A.x + Bc.x), silu_f(xq.y * A.y + Bc.y), silu_f(xq.z * A.z + Bc.z), silu_f(xq.w * A.w + Bc.w));
                    if constexpr (AFF == 2) { xq.x += E.x; xq.y += E.y; xq.z += E.z; xq.w += E.w; }
                }
                if (!(((vmask >> (4 + j)) & 1) && xok[t])) xq = zero4;           // zero padding applies to the transformed tensor
                uint2 xl; const uint2 xh = split_quad(xq, xl);
                const int xo = pl * PX + ((t * 128 + quad * 8) ^ xswz_w);
                *reinterpret_cast<uint2*>(smem + XHI + xo) = xh;
                *reinterpret_cast<uint2*>(smem + XLO + xo) = xl;
            }
        }
        if (!ONE && tid < 32) {
            const float4 gq = (vmask >> 8) & 1 ? rGh : zero4;
            uint2 lo; const uint2 hi = split_quad(gq, lo);
            const int row = tid < 16 ? 0 : 65;
            *reinterpret_cast<uint2*>(smem + GHI + row * PG + quad * 8) = hi;
            *reinterpret_cast<uint2*>(smem + GLO + row * PG + quad * 8) = lo;
        }
    };

    // ---- fragment addresses: lane 4q + p of 16-lane group gq supplies (pixel row q, channels 4p .. 4p + 3) of its group's block;
    // group = (channel half = bit 0, k half = bit 1)
    const int grp = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int prow = 8 * (grp >> 1) + q4;                                        // + 16 ks (+ 4 for the second block)
    int xoff[TMW];                                                               // the pixel's low bits are q4's (every other term is a multiple of 4)
#pragma unroll
    for (int t = 0; t < TMW; ++t)
        xoff[t] = prow * PX + ((((wm * TMW + t) * 32 + 16 * (grp & 1) + 4 * p4) * 2) ^ ((TMW == 2 ? (q4 & 3) : ((q4 >> 1) & 1)) << 6));
    const int goff = (prow + 1) * PG + (wn * 32 + 16 * (grp & 1) + 4 * p4) * 2;   // G pixel k sits in row k + 1

    f32x16 acc[TMW][NDX];
#pragma unroll
    for (int t = 0; t < TMW; ++t)
#pragma unroll
        for (int x = 0; x < NDX; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][x][r] = 0.f;

#ifdef HD_STAMPS
    unsigned long long c_store = 0, c_bar = 0, c_mma = 0, t0, t1;
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
    request(0);
    for (int s = 0; s < nslices; ++s) {
#ifdef HD_STAMPS
        t0 = __builtin_readcyclecounter();
#endif
        store();
#ifdef HD_STAMPS
        t1 = __builtin_readcyclecounter(); c_store += t1 - t0;
#endif
        __syncthreads();
#ifdef HD_STAMPS
        t0 = __builtin_readcyclecounter(); c_bar += t0 - t1;
#endif
        request(min(s + 1, nslices - 1));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 ah[TMW], al[TMW];
#pragma unroll
            for (int t = 0; t < TMW; ++t) {
                ah[t] = tr_read8(smem + XHI + xoff[t] + ks * 16 * PX, 4 * PX);
                al[t] = tr_read8(smem + XLO + xoff[t] + ks * 16 * PX, 4 * PX);
            }
#pragma unroll
            for (int dx = 0; dx < NDX; ++dx) {
                // kx = dx: G[k + 1 - kx] (the centre tap only for a 1 x 1 filter)
                const int sh = ONE ? 0 : (1 - dx) * PG;
                const bf16x8 bh = tr_read8(smem + GHI + goff + ks * 16 * PG + sh, 4 * PG);
                const bf16x8 bl = tr_read8(smem + GLO + goff + ks * 16 * PG + sh, 4 * PG);
#pragma unroll
                for (int t = 0; t < TMW; ++t) {
                    acc[t][dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh, acc[t][dx], 0, 0, 0);
                    acc[t][dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl, acc[t][dx], 0, 0, 0);
                    acc[t][dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh, acc[t][dx], 0, 0, 0);
                }
            }
        }
#ifdef HD_STAMPS
        t1 = __builtin_readcyclecounter(); c_mma += t1 - t0;
#endif
        __syncthreads();
#ifdef HD_STAMPS
        c_bar += __builtin_readcyclecounter() - t1;
#endif
    }
#ifdef HD_STAMPS
    if (tid == 0 && blockIdx.x < 1024 && blockIdx.y == 0) {
        g_wgd_stamps[blockIdx.x][0] = c_store; g_wgd_stamps[blockIdx.x][1] = c_bar; g_wgd_stamps[blockIdx.x][2] = c_mma;
        g_wgd_stamps[blockIdx.x][3] = __builtin_readcyclecounter() - t_begin;
    }
#endif
    const int N = NDX * a.F;
    float* out = a.partial + ((size_t)(split * NDX + (ONE ? 0 : dyi)) * a.M) * N;
#pragma unroll
    for (int t = 0; t < TMW; ++t)
#pragma unroll
        for (int dx = 0; dx < NDX; ++dx)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * BM + (wm * TMW + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int n = dx * a.F + nt * 64 + wn * 32 + l31;
                out[(size_t)m * N + n] = acc[t][dx][r];
            }
    if (do_bias) {                                                               // uniform per workgroup
        float* red = reinterpret_cast<float*>(smem);                             // [16 pixel groups][64 channels]; the images are dead
        *reinterpret_cast<float4*>(red + pg * 64 + quad * 4) = bsum;
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
            for (int k = 0; k < 16; ++k) t += red[k * 64 + tid];
            a.biaspart[(size_t)split * a.F + nt * 64 + tid] = t;
        }
    }
}

// ---- all nine taps in one workgroup over a rolling window of activation positions ------------------------------------------------
// wgrad_direct_kernel gives each row tap its own workgroup: the gradient slice is staged three times and every activation position
// three times (once per row tap that touches it) -- 49 KB of operands per 72 MFMAs of a wave.  Here a workgroup of EIGHT waves (one per
// CU: 128 input channels x 64 output channels x 9 taps, 9 accumulator tiles per wave) keeps the activation positions
// [k0 - (W+1), k0 + 63 + (W+1)] of the padded position space resident in an LDS ring (R = 128 or 256 positions, index = position & (R-1))
// and adds the 64 new positions of each slice: in the padded space the row tap is the position offset (ky-1)(W+1) -- the zero row
// between samples and the zero column between rows make every out-of-map neighbour a stored zero -- so the A fragments of the three
// row taps are three transposed reads of the same ring, and each activation is loaded from memory ONCE per workgroup: 49 KB per
// 108 MFMAs of each of 8 waves.  Column taps, fragment reads, splitting, bias sums and the partial layout are wgrad_direct_kernel's.
// AFF as there (0 / 1 / 2).  Layers with more than 64 input channels, 3 x 3 filters, plain source addressing.
// PLAIN: one bf16 MFMA per product (the optional bf16 training arithmetic): no lo images, a third of the MFMAs.
template <int AFF, bool PLAIN = false>
__global__ __launch_bounds__(512) void wgrad_direct9_kernel(WgdArgs a, int R) {
    constexpr int PX = 256, PG = 64 * 2 + 64;                                    // bytes per position: 128 channels (swizzled) / 64 channels + pad
    extern __shared__ __attribute__((aligned(16))) char smem9[];
    char* const XHI = smem9; char* const XLO = XHI + (size_t)R * PX;
    char* const GHI = XLO + (size_t)R * PX; char* const GLO = GHI + 66 * PG;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = w >> 1, wn = w & 1;                                           // 4 x 2 waves of 32 x 32 (x 9 taps)
    const int per = a.Mtiles * a.Ntiles;
    int split, inner;
    if ((a.nsplit & 7) == 0 && a.xcd_group) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; split = xcd + 8 * (j / per); inner = j % per; }
    else { split = blockIdx.x / per; inner = blockIdx.x % per; }
    const int nt = inner % a.Ntiles, mt = inner / a.Ntiles;
    const int Wp = a.W + 1, Hp = a.H + 1, Ktot = a.B * Hp * Wp, Cin = a.C0 + a.C1, RM = R - 1;
    const int s0 = split * a.slices_per_split;
    const int nslices = min(a.slices_per_split, a.total_slices - s0);
    const bool do_bias = a.biaspart != nullptr && mt == 0;

    // ---- loader roles.  X: positions xp0 + 16 j (j < 4) of a 64-position chunk, channel quad xq (32 quads = 128 channels);
    // G: positions gp0 + 32 j (j < 2), channel quad gq (16 quads); threads 0..31 also carry the two halo positions of G.
    const int xp0 = tid >> 5, xq = tid & 31, gp0 = tid >> 4, gq = tid & 15;
    const int gch = nt * 64 + gq * 4;
    const int xc = mt * 128 + xq * 4;
    const bool xok = xc < Cin;
    const float* xsrc; int xC;
    if (xc < a.C0) { xsrc = a.x0 + xc; xC = a.C0; } else { xsrc = a.x1 + (xc - a.C0); xC = a.C1; }
    if (!xok) { xsrc = a.x0; xC = a.C0; }
    auto locate = [&](int kp, int& pix, int& smp) {                               // padded position -> source pixel (or -1) and sample
        const bool in = kp >= 0 && kp < Ktot;
        const unsigned k = in ? (unsigned)kp : 0u;
        const unsigned yq = __umulhi(k, a.magicW), xp = k - yq * Wp, b = __umulhi(yq, a.magicH), yp = yq - b * Hp;
        pix = in && (int)xp < a.W && (int)yp < a.H ? ((int)b * a.H + (int)yp) * a.W + (int)xp : -1;
        smp = in ? (int)b : 0;
    };
    struct Coef { float4 A, B, E; };
    auto coef = [&](int smp) __attribute__((always_inline)) {
        const size_t co = (size_t)smp * a.aff_bs + (xok ? xc : 0);
        Coef c;
        c.A = a.affA ? *reinterpret_cast<const float4*>(a.affA + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        c.A.x += a.aff_addA; c.A.y += a.aff_addA; c.A.z += a.aff_addA; c.A.w += a.aff_addA;
        c.B = *reinterpret_cast<const float4*>(a.affB + co);
        c.E = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (AFF == 2) c.E = *reinterpret_cast<const float4*>(a.affE + co);
        return c;
    };
    float4 rX[4], rG[2], rGh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cA = make_float4(1.f, 1.f, 1.f, 1.f), cB = make_float4(0.f, 0.f, 0.f, 0.f), cE = cB;
    int s_first = 0; unsigned sdiff = 0, vmask = 0;                              // vmask bits 0-3: X item valid; 4-5: G item; 6: halo
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bsum = zero4;
    // X chunk = the 64 positions starting at P; request / write are separate so that a chunk travels in registers under the MFMAs
    auto x_request = [&](int P) {
        vmask &= ~0xfu; sdiff = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int pix, smp;
            locate(P + xp0 + 16 * j, pix, smp);
            rX[j] = *reinterpret_cast<const float4*>(xsrc + (size_t)(pix < 0 ? 0 : pix) * xC);
            vmask |= (pix >= 0 ? 1u : 0u) << j;
            if constexpr (AFF != 0) {
                if (j == 0) { s_first = smp; const Coef c = coef(smp); cA = c.A; cB = c.B; cE = c.E; }
                else sdiff |= (unsigned)min(max(smp - s_first, 0), 255) << (8 * j);
            }
        }
    };
    auto x_write = [&](int P) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = P + xp0 + 16 * j;
            float4 xq4 = rX[j];
            if constexpr (AFF != 0) {
                float4 A = cA, Bc = cB, E = cE;
                const int dsm = (int)((sdiff >> (8 * j)) & 255u);
                if (dsm != 0) { const Coef c = coef(s_first + dsm); A = c.A; Bc = c.B; E = c.E; }
                xq4 = make_float4(silu_rcp(xq4.x * A.x + Bc.x), silu_rcp(xq4.y * A.y + Bc.y), silu_rcp(xq4.z * A.z + Bc.z), silu_rcp(xq4.w * A.w + Bc.w));
                if constexpr (AFF == 2) { xq4.x += E.x; xq4.y += E.y; xq4.z += E.z; xq4.w += E.w; }
            }
            if (!(((vmask >> j) & 1) && xok)) xq4 = zero4;
            uint2 lo; const uint2 hi = split_quad(xq4, lo);
            const int o = (pos & RM) * PX + ((xq * 8) ^ ((pos & 3) << 6));
            *reinterpret_cast<uint2*>(XHI + o) = hi;
            if constexpr (!PLAIN) *reinterpret_cast<uint2*>(XLO + o) = lo;
        }
    };
    auto g_request = [&](int k0) {
        vmask &= 0xfu;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int pix, smp;
            locate(k0 + gp0 + 32 * j, pix, smp);
            rG[j] = *reinterpret_cast<const float4*>(a.g + (size_t)(pix < 0 ? 0 : pix) * a.F + gch);
            vmask |= (pix >= 0 ? 1u : 0u) << (4 + j);
        }
        if (tid < 32) {
            int pix, smp;
            locate(tid < 16 ? k0 - 1 : k0 + 64, pix, smp);
            rGh = *reinterpret_cast<const float4*>(a.g + (size_t)(pix < 0 ? 0 : pix) * a.F + gch);
            vmask |= (pix >= 0 ? 1u : 0u) << 6;
        }
    };
    auto g_write = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 g4 = (vmask >> (4 + j)) & 1 ? rG[j] : zero4;
            if (do_bias) { bsum.x += g4.x; bsum.y += g4.y; bsum.z += g4.z; bsum.w += g4.w; }
            uint2 lo; const uint2 hi = split_quad(g4, lo);
            const int o = (gp0 + 32 * j + 1) * PG + gq * 8;
            *reinterpret_cast<uint2*>(GHI + o) = hi;
            if constexpr (!PLAIN) *reinterpret_cast<uint2*>(GLO + o) = lo;
        }
        if (tid < 32) {
            const float4 g4 = (vmask >> 6) & 1 ? rGh : zero4;
            uint2 lo; const uint2 hi = split_quad(g4, lo);
            const int o = (tid < 16 ? 0 : 65) * PG + gq * 8;
            *reinterpret_cast<uint2*>(GHI + o) = hi;
            if constexpr (!PLAIN) *reinterpret_cast<uint2*>(GLO + o) = lo;
        }
    };

    // ---- fragment addresses (lane 4q + p of 16-lane group: position row q, channels 4p .. 4p+3 of its block)
    const int grp = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int prow = 8 * (grp >> 1) + q4;
    const int xchb = (wm * 32 + 16 * (grp & 1) + 4 * p4) * 2;                    // byte offset of this lane's channels in an X position row
    const int goff = (prow + 1) * PG + (wn * 32 + 16 * (grp & 1) + 4 * p4) * 2;

    f32x16 acc[3][3];
#pragma unroll
    for (int y = 0; y < 3; ++y)
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[y][x][r] = 0.f;

    // ---- prologue: the ring holds [k0 - Wp, k0 + Wp) before slice 0 adds [k0 + Wp, k0 + Wp + 64)
    const int kbeg = s0 * 64;
    {
        const int nlead = (2 * Wp + 63) >> 6;                                    // whole chunks that cover the 2 Wp leading positions
        for (int c = nlead; c >= 1; --c) {
            x_request(kbeg + Wp - 64 * c);
            x_write(kbeg + Wp - 64 * c);
        }
    }
    x_request(kbeg + Wp);
    g_request(kbeg);
#ifdef HD_STAMPS
    unsigned long long c_store = 0, c_bar = 0, c_req = 0, c_mma = 0, t0, t1;
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < nslices; ++s) {
        const int k0 = kbeg + 64 * s;
#ifdef HD_STAMPS
        t0 = __builtin_readcyclecounter();
#endif
        x_write(k0 + Wp);
        g_write();
#ifdef HD_STAMPS
        t1 = __builtin_readcyclecounter(); c_store += t1 - t0;
#endif
        __syncthreads();
#ifdef HD_STAMPS
        t0 = __builtin_readcyclecounter(); c_bar += t0 - t1;
#endif
        { const int sn = min(s + 1, nslices - 1); x_request(kbeg + 64 * sn + Wp); g_request(kbeg + 64 * sn); }
#ifdef HD_STAMPS
        t1 = __builtin_readcyclecounter(); c_req += t1 - t0;
#endif
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 bh[3], bl[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {                                     // kx = dx: G[k + 1 - kx]
                bh[dx] = tr_read8(GHI + goff + ks * 16 * PG + (1 - dx) * PG, 4 * PG);
                if constexpr (!PLAIN) bl[dx] = tr_read8(GLO + goff + ks * 16 * PG + (1 - dx) * PG, 4 * PG);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {                                     // ky = dy: X[k + (ky - 1) Wp], two 4-position blocks
                const int p0 = k0 + (dy - 1) * Wp + ks * 16 + prow, p1 = p0 + 4;
                const int o0 = (p0 & RM) * PX + (xchb ^ ((p0 & 3) << 6)), o1 = (p1 & RM) * PX + (xchb ^ ((p1 & 3) << 6));
                const bf16x8 ah = tr_read8(XHI + o0, o1 - o0);
                if constexpr (PLAIN) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) acc[dy][dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[dx], acc[dy][dx], 0, 0, 0);
                } else {
                    const bf16x8 al = tr_read8(XLO + o0, o1 - o0);
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        acc[dy][dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[dx], acc[dy][dx], 0, 0, 0);
                        acc[dy][dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[dx], acc[dy][dx], 0, 0, 0);
                        acc[dy][dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[dx], acc[dy][dx], 0, 0, 0);
                    }
                }
            }
        }
#ifdef HD_STAMPS
        t0 = __builtin_readcyclecounter(); c_mma += t0 - t1;
#endif
        __syncthreads();
#ifdef HD_STAMPS
        c_bar += __builtin_readcyclecounter() - t0;
#endif
    }
#ifdef HD_STAMPS
    if (tid == 0 && blockIdx.x < 1024) {
        g_wgd_stamps[blockIdx.x][0] = c_store; g_wgd_stamps[blockIdx.x][1] = c_bar; g_wgd_stamps[blockIdx.x][2] = c_mma;
        g_wgd_stamps[blockIdx.x][3] = __builtin_readcyclecounter() - t_begin; g_wgd_stamps[blockIdx.x][4] = c_req;
    }
#endif
    const int N = 3 * a.F;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        float* out = a.partial + ((size_t)(split * 3 + dy) * a.M) * N;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * 128 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int n = dx * a.F + nt * 64 + wn * 32 + l31;
                out[(size_t)m * N + n] = acc[dy][dx][r];
            }
    }
    if (do_bias) {                                                               // uniform per workgroup
        float* red = reinterpret_cast<float*>(smem9);                           // [32 position groups][64 channels]
        *reinterpret_cast<float4*>(red + gp0 * 64 + gq * 4) = bsum;
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
            for (int k = 0; k < 32; ++k) t += red[k * 64 + tid];
            a.biaspart[(size_t)split * a.F + nt * 64 + tid] = t;
        }
    }
}

// dW[co][ci][ky][kx] (torch layout, KT x KT taps, KT = 3 or 1) (+)= scale * sum_split partial[split][ky][ci (< Mpad)][kx*Cout + co]
__global__ __launch_bounds__(256) void wg_reduce_kernel(const float* __restrict__ partial, int nsplit, int Cin, int Mpad, int Cout, int KT, float scale,
                                                        int accumulate, float* __restrict__ dW, const float* __restrict__ biaspart = nullptr,
                                                        float* __restrict__ db = nullptr, int nmain = 0) {
    if (db && (int)blockIdx.x >= nmain) {                              // trailing blocks: db[co] = sum_split biaspart[split][co]
        const int co = ((int)blockIdx.x - nmain) * 256 + threadIdx.x;
        if (co >= Cout) return;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int sp = 0;
        for (; sp + 8 <= nsplit; sp += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += biaspart[(size_t)(sp + j) * Cout + co];
        }
        for (; sp < nsplit; ++sp) a[sp & 7] += biaspart[(size_t)sp * Cout + co];
        db[co] = (accumulate ? db[co] : 0.f) + scale * (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
        return;
    }
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // over [KT][Mpad][KT*Cout]
    const size_t per = (size_t)KT * Mpad * KT * Cout;
    if (i >= per) return;
    const int n = (int)(i % ((size_t)KT * Cout)), ci = (int)((i / ((size_t)KT * Cout)) % Mpad), ky = (int)(i / ((size_t)KT * Cout * Mpad));
    if (ci >= Cin) return;                                             // zero rows that pad the activation image to 128
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // independent chains: see sum_rows_kernel
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += partial[(size_t)(sp + j) * per + i];
    }
    for (; sp < nsplit; ++sp) a[sp & 7] += partial[(size_t)sp * per + i];
    const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    const int kx = n / Cout, co = n % Cout;
    float* d = dW + (((size_t)co * Cin + ci) * KT + ky) * KT + kx;
    *d = (accumulate ? *d : 0.f) + scale * s;
}

// data-gradient weight: Wt[ci][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx]  (both torch layout; KK = K*K taps, 9 or 1)
__global__ __launch_bounds__(256) void flip_weight_kernel(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ wt, int KK = 9) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Cout * Cin * KK) return;
    const int tap = (int)(i % KK), co = (int)((i / KK) % Cout), ci = (int)(i / ((size_t)KK * Cout));
    wt[i] = w[((size_t)co * Cin + ci) * KK + (KK - 1 - tap)];
}

// ---- elementwise ---------------------------------------------------------------------------------------------------
// a = silu(u * (scale + 1) + shift), film = [B][2C] (scale | shift)
__global__ __launch_bounds__(256) void film_silu_fwd_kernel(const float* __restrict__ u, const float* __restrict__ film, int film_bs, int HW, int C,
                                                            size_t n4, float* __restrict__ a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t e = i * 4;
    const int c = (int)(e % C), b = (int)(e / ((size_t)HW * C));
    const float4 v = *reinterpret_cast<const float4*>(u + e);
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f);            // SR3 (film_bs == C): additive only, src/model/hicedrn_sr3_Diff.py:254-265
    if (film_bs == 2 * C) sc = *reinterpret_cast<const float4*>(film + (size_t)b * film_bs + c);
    const float4 sh = *reinterpret_cast<const float4*>(film + (size_t)b * film_bs + (film_bs - C) + c);
    float4 o;
    o.x = silu_f(v.x * (sc.x + 1.f) + sh.x); o.y = silu_f(v.y * (sc.y + 1.f) + sh.y);
    o.z = silu_f(v.z * (sc.z + 1.f) + sh.z); o.w = silu_f(v.w * (sc.w + 1.f) + sh.w);
    *reinterpret_cast<float4*>(a + e) = o;
}

// g (in: dL/da * 1/gscale; out: dL/du), u: pre-FiLM conv output.  v = u (sc+1) + sh; dv = gscale * g * silu'(v);
// du = dv (sc+1); part[(b*nchunk + chunk)][0][c] = sum dv*u (d scale), [1][c] = sum dv (d shift).
__global__ __launch_bounds__(256) void film_silu_bwd_kernel(float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ film,
                                                            int film_bs, int HW, int C, int chunk, float gscale, float* __restrict__ part) {
    const int b = blockIdx.y, ck = blockIdx.x, nchunk = gridDim.x;
    const int p0 = ck * chunk, p1 = min(HW, p0 + chunk);
    for (int c = threadIdx.x; c < C; c += 256) {
        const bool two = film_bs == 2 * C;
        const float sc = two ? film[(size_t)b * film_bs + c] + 1.f : 1.f, sh = film[(size_t)b * film_bs + (film_bs - C) + c];
        float s_scale = 0.f, s_shift = 0.f;
        for (int p = p0; p < p1; ++p) {
            const size_t e = ((size_t)b * HW + p) * C + c;
            const float uu = u[e];
            const float v = uu * sc + sh;
            const float sg = sigmoid_f(v);
            const float dv = gscale * g[e] * (sg * (1.f + v * (1.f - sg)));
            s_scale += dv * uu; s_shift += dv;
            g[e] = dv * sc;
        }
        float* d = part + (size_t)(b * nchunk + ck) * film_bs;      // [d scale | d shift], or [d shift] alone
        if (two) d[c] = s_scale;
        d[(film_bs - C) + c] = s_shift;
    }
}

// d film row of one block from the convolution epilogue's per-tile sums (EP_FILM_SILU_BWD): part [B][slots][C][2] = (sum dv u, sum dv) ->
// out [B][FW] = [d scale | d shift] (FW == C: d shift alone), slots added in order
__global__ __launch_bounds__(256) void film_part_reduce_kernel(const float* __restrict__ part, int slots, int C, int FW, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    float a = 0.f, q = 0.f;
    for (int s = 0; s < slots; ++s) {
        const float2 v = *reinterpret_cast<const float2*>(part + (((size_t)b * slots + s) * C + c) * 2);
        a += v.x; q += v.y;
    }
    if (FW == 2 * C) out[(size_t)b * FW + c] = a;
    out[(size_t)b * FW + (FW - C) + c] = q;
}

// out[batch][col] (+)= scale * sum_row in[batch][row][col]
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ in, int nrows, int ncols, float scale, int accumulate,
                                                       float* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x, bt = blockIdx.y;
    if (col >= ncols) return;
    const float* p = in + (size_t)bt * nrows * ncols + col;
    // eight independent partial sums: the loads of a serial chain would each wait out the full memory latency
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = 0;
    for (; r + 8 <= nrows; r += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += p[(size_t)(r + j) * ncols];
    }
    for (; r < nrows; ++r) a[r & 7] += p[(size_t)r * ncols];
    const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    float* d = out + (size_t)bt * ncols + col;
    *d = (accumulate ? *d : 0.f) + scale * s;
}

// ---- small dense layers (time MLP, FiLM projections); weights in torch layout W[N][K] ---------------------------------
// act: 0 none, 1 SiLU, 2 exact GELU applied to X on the way in
__device__ __forceinline__ float act_f(float x, int act) {
    return act == 1 ? silu_f(x) : act == 2 ? 0.5f * x * (1.f + erff(x * 0.70710678118654752f)) : x;
}
__device__ __forceinline__ float dact_f(float x, int act) {
    if (act == 1) { const float s = sigmoid_f(x); return s * (1.f + x * (1.f - s)); }
    if (act == 2) return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    return 1.f;
}

// Y[b][n] = bias[n] + sum_k act(X[b][k]) * W[n][k]; one wave per output column n, lanes over k
// grid.y = layer index (the per-block FiLM projections run as one launch): W, bias, Y advance by wstride, bstride, ystride floats
__global__ __launch_bounds__(256) void lin_fwd_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, const float* __restrict__ bias,
                                                      int Bn, int K, int N, int act, float* __restrict__ Y, int ldy, size_t wstride, size_t bstride,
                                                      size_t ystride) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    W += blockIdx.y * wstride; bias += blockIdx.y * bstride; Y += blockIdx.y * ystride;
    for (int b = 0; b < Bn; ++b) {
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += act_f(X[(size_t)b * ldx + k], act) * W[(size_t)n * K + k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) Y[(size_t)b * ldy + n] = s + bias[n];
    }
}

// The activated input of the per-block FiLM projections, once per step instead of once per output column: A[b][k] = act(X[b][k]) and its
// transpose AT[k][b] (Bp = the batch padded to whole waves; padding columns are zero).
__global__ __launch_bounds__(256) void act_rows_kernel(const float* __restrict__ X, int Bn, int K, int act, int Bp, float* __restrict__ A, float* __restrict__ AT) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Bp * K) return;
    const int b = (int)(i / K), k = (int)(i % K);
    const float v = b < Bn ? act_f(X[(size_t)b * K + k], act) : 0.f;
    if (b < Bn) A[i] = v;
    AT[(size_t)k * Bp + b] = v;
}

// Y[b][n] = bias[n] + sum_k AT[k][b] W[n][k] with one LANE per sample: the wave walks k once for four output columns, its loads of AT are one
// coalesced row per k and the four weights of a k are wave-uniform.  (lin_fwd_kernel spends a wave per column and sample, with a shuffle
// reduction and an activation per term: 1.5 ms for hicedrn's 32 x 512 FiLM columns at 64 samples, 89 % of it exp.)  grid = (N / 4 / 4, layers, Bp / 64).
__global__ __launch_bounds__(256) void lin_fwd_t_kernel(const float* __restrict__ AT, int Bp, const float* __restrict__ W, const float* __restrict__ bias, int Bn, int K,
                                                        int N, float* __restrict__ Y, int ldy, size_t wstride, size_t bstride, size_t ystride) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 4 + wv) * 4, b = blockIdx.z * 64 + lane;
    if (n0 >= N) return;
    W += blockIdx.y * wstride + (size_t)n0 * K; bias += blockIdx.y * bstride; Y += blockIdx.y * ystride;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* at = AT + b;
    for (int k = 0; k < K; k += 4) {
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = at[(size_t)(k + u) * Bp];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n0 + j < N) {
                const float4 w = *reinterpret_cast<const float4*>(W + (size_t)j * K + k);      // wave-uniform address
                acc[j] += x[0] * w.x + x[1] * w.y + x[2] * w.z + x[3] * w.w;
            }
        }
    }
    if (b < Bn)
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n0 + j < N) Y[(size_t)b * ldy + n0 + j] = acc[j] + bias[n0 + j];
}

// dW[n][k] = sum_b dY[b][n] * act(X[b][k]);  db[n] = sum_b dY[b][n]
__global__ __launch_bounds__(256) void lin_bwd_w_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, int Bn, int K, int N,
                                                        int act, float* __restrict__ dW, float* __restrict__ db, size_t dystride, size_t wstride, size_t bstride) {
    const int n = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    dY += blockIdx.z * dystride; dW += blockIdx.z * wstride; db += blockIdx.z * bstride;      // grid.z = layer index
    if (k < K) {
        float s = 0.f;
        for (int b = 0; b < Bn; ++b) s += dY[(size_t)b * ldy + n] * act_f(X[(size_t)b * ldx + k], act);
        dW[(size_t)n * K + k] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < Bn; ++b) s += dY[(size_t)b * ldy + n];
        db[n] = s;
    }
}

// dX[b][k] (+)= (sum_n dY[b][n] * W[n][k]) * act'(Xpre[b][k])   (act' only when Xpre is given)
__global__ __launch_bounds__(256) void lin_bwd_x_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ W, int Bn, int K, int N,
                                                        const float* __restrict__ Xpre, int ldx, int act, int accumulate, float* __restrict__ dX, int lddx,
                                                        size_t dystride, size_t wstride, size_t xstride) {
    const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    dY += blockIdx.z * dystride; W += blockIdx.z * wstride; dX += blockIdx.z * xstride;       // grid.z = layer index (per-layer partials)
    float a4[4] = {0.f, 0.f, 0.f, 0.f};                     // independent chains (see sum_rows_kernel)
    int n = 0;
    for (; n + 4 <= N; n += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a4[j] += dY[(size_t)b * ldy + n + j] * W[(size_t)(n + j) * K + k];
    }
    for (; n < N; ++n) a4[0] += dY[(size_t)b * ldy + n] * W[(size_t)n * K + k];
    float s = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    if (Xpre) s *= dact_f(Xpre[(size_t)b * ldx + k], act);
    float* d = dX + (size_t)b * lddx + k;
    *d = (accumulate ? *d : 0.f) + s;
}

// x *= act'(pre) elementwise (used to turn d silu(t) into d t)
__global__ __launch_bounds__(256) void dact_mul_kernel(float* __restrict__ x, const float* __restrict__ pre, size_t n, int act) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] *= dact_f(pre[i], act);
}

// SinusoidalPosEmb (src/model/hicedrn_Diff.py:122-134): emb[b] = [sin(t f_k), cos(t f_k)], f_k = exp(-k ln(1e4)/(half-1))
// SR3 PositionalEncoding of the continuous noise level (src/hicdiff_sr3.py:155-165): f_k = exp(-ln(1e4) k / half)
__global__ __launch_bounds__(256) void sin_emb_kernel(const void* __restrict__ t, int t_float, int sr3, int dim, float* __restrict__ emb) {
    const int b = blockIdx.x, half = dim / 2;
    const float tv = t_float ? reinterpret_cast<const float*>(t)[b] : (float)reinterpret_cast<const long long*>(t)[b];
    for (int i = threadIdx.x; i < dim; i += 256) {
        const int k = i < half ? i : i - half;
        const float a = tv * (sr3 ? expf(-9.210340371976184f * ((float)k / (float)half)) : expf((float)k * -(9.210340371976184f / (float)(half - 1))));
        emb[(size_t)b * dim + i] = i < half ? sinf(a) : cosf(a);
    }
}

// ---- first / last convolution weight gradients (1-2 channels on one side) -----------------------------------------------
// part[(b*nrb + rb)][c][j*KS*KS + tap] = sum over the row block of big[p][c] * small_j[p + sign*(tap offset)]
// KS = 3 (hicedrn head / tail) or 7 (the UNet's init_conv, src/hicdiff.py:279).
// NG: threads per channel, each taking every NG-th tap (the UNet's first convolution has 64 output channels and 49 taps: 4 x 13 taps per channel
// keep the workgroup's 256 threads busy); C * NG <= 256 or NG == 1.
template <int KS, int NG>
__global__ __launch_bounds__(256) void small_conv_wgrad_kernel(const float* __restrict__ big, const float* __restrict__ s0, const float* __restrict__ s1,
                                                               int J, int S, int C, int RB, int sign, float* __restrict__ part) {
    constexpr int R = KS / 2, T = KS * KS, TP = (T + NG - 1) / NG;
    extern __shared__ float sm[];                           // [J][RB + 2R][S + 2R]
    const int b = blockIdx.y, rb = blockIdx.x, nrb = gridDim.x, y0 = rb * RB, LW = S + 2 * R, LH = RB + 2 * R;
    for (int i = threadIdx.x; i < J * LH * LW; i += 256) {
        const int j = i / (LH * LW), r = (i / LW) % LH, x = i % LW;
        const int yy = y0 + r - R, xx = x - R;
        const float* src = j == 0 ? s0 : s1;
        sm[i] = (yy >= 0 && yy < S && xx >= 0 && xx < S) ? src[((size_t)b * S + yy) * S + xx] : 0.f;
    }
    __syncthreads();
    const int tg = NG == 1 ? 0 : (int)threadIdx.x / C;
    if (NG > 1 && tg >= NG) return;
    for (int c = NG == 1 ? threadIdx.x : (int)threadIdx.x % C; c < C; c += NG == 1 ? 256 : C) {
        for (int j = 0; j < J; ++j) {                       // one small plane at a time: TP accumulators per thread
            float acc[TP];
            int off[TP];
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                const int t = min(tg + i * NG, T - 1);       // (a clamped duplicate tap is computed and dropped below)
                acc[i] = 0.f;
                off[i] = (j * LH + R + sign * (t / KS - R)) * LW + R + sign * (t % KS - R);
            }
            // eight pixels' values requested together: one load per pixel in flight at a time made this a chain of exposed HBM round trips
            // (the UNet's first convolution: 324 us for 67 MB).  Same summation order as the scalar loop.
            for (int yy = 0; yy < RB && y0 + yy < S; ++yy)
                for (int x0 = 0; x0 < S; x0 += 8) {
                    const float* src = big + (((size_t)b * S + y0 + yy) * S + x0) * C + c;
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)min(u, S - 1 - x0) * C];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (x0 + u < S) {
                            const float* row = sm + yy * LW + x0 + u;
#pragma unroll
                            for (int i = 0; i < TP; ++i) acc[i] += v[u] * row[off[i]];
                        }
                    }
                }
            float* d = part + ((size_t)(b * nrb + rb) * C + c) * (J * T) + j * T;
#pragma unroll
            for (int i = 0; i < TP; ++i) { const int t = tg + i * NG; if (t < T) d[t] = acc[i]; }
        }
        if (NG > 1) break;
    }
}

// ---- the UNet's 1x1 final convolution (C -> 1, src/hicdiff.py:319): out[p] = sum_c x[p][c] w[c] + b ----------------------------------
// part[block][c] = sum over the block's pixels of x[p][c] * dout[p] (d w); dx[p][c] = dout[p] * w[c]
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ w, size_t P, int C,
                                                         int rows, float* __restrict__ dx, float* __restrict__ part) {
    const size_t p0 = (size_t)blockIdx.x * rows;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float wc = w[c];
        float s = 0.f;
        for (int r = 0; r < rows && p0 + r < P; ++r) {
            const float g = dout[p0 + r];
            s += x[(p0 + r) * C + c] * g;
            dx[(p0 + r) * C + c] = g * wc;
        }
        part[(size_t)blockIdx.x * C + c] = s;
    }
}

// ---- loss ---------------------------------------------------------------------------------------------------------
// per[b] = w_b mean_i f(out - eps); dout = w_b f'(out - eps) / (B * n)   (l2: f = d^2; l1: f = |d|; w_b = p2_loss_weight[t_b], src/hicdiff.py:746,
// or 1 when lw == nullptr)
// per[b] = per-sample loss, per[B + b] = sum of the sample's dout (the last convolution's bias gradient)
// objective (src/hicdiff.py:733-741): 0 the target is the noise; 1 x_start itself; 2 v = a_t eps - s_t x_start (predict_v, :542-546)
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ out, const float* __restrict__ eps, int n, int B, int l2,
                                                        float* __restrict__ per, float* __restrict__ dout, int objective = 0,
                                                        const float* __restrict__ x0 = nullptr, const float* __restrict__ a_t = nullptr,
                                                        const float* __restrict__ s_t = nullptr, const float* __restrict__ lw = nullptr) {
    __shared__ float red[256], red2[256];
    const int b = blockIdx.x;
    const float wb = lw ? lw[b] : 1.f;
    const float inv = wb / ((float)B * (float)n);
    const float av = objective == 2 ? a_t[b] : 0.f, sv = objective == 2 ? s_t[b] : 0.f;
    float s = 0.f, sg = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const size_t e = (size_t)b * n + i;
        const float target = objective == 0 ? eps[e] : objective == 1 ? x0[e] : av * eps[e] - sv * x0[e];
        const float d = out[e] - target;
        s += l2 ? d * d : fabsf(d);
        const float g = l2 ? 2.f * d * inv : (d > 0.f ? inv : d < 0.f ? -inv : 0.f);
        dout[(size_t)b * n + i] = g;
        sg += g;
    }
    red[threadIdx.x] = s; red2[threadIdx.x] = sg;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) { red[threadIdx.x] += red[threadIdx.x + m]; red2[threadIdx.x] += red2[threadIdx.x + m]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { per[b] = wb * (red[0] / (float)n); per[B + b] = red2[0]; }
}
__global__ void mean_kernel(const float* __restrict__ per, int B, float* __restrict__ loss, float* __restrict__ dbias) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f, g = 0.f;
        for (int b = 0; b < B; ++b) { s += per[b]; g += per[B + b]; }
        *loss = s / (float)B; *dbias = g;
    }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n4, float* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(o)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

// torch.optim.Adam (no weight decay, no amsgrad): train.py:111
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   float b1, float b2, float eps, float step_size, float inv_sqrt_c2, float gscale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gg = g[i] * gscale;
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm; v[i] = vv;
    p[i] -= step_size * mm / (sqrtf(vv) * inv_sqrt_c2 + eps);
}

}  // namespace

// =====================================================================================================================
// Weight gradient of a KT x KT (3 or 1), stride-1, same-padded convolution as a reusable component: one instance per map geometry
// (B, H, W), because the zero padding of its operand images is positional.  rewrite() fills the activation image (gside == false;
// channel-concatenated inputs are two calls with row0 = 0 and row0 = C0) or the gradient image; run() multiplies and reduces.
struct Wgrad {
    int B = 0, H = 0, W = 0, P = 0, maxCin = 0, maxCout = 0;
    bool narrow = false;
    size_t Kpad = 0, guard = 0, ld = 0, partial_floats = 0;
    unsigned short *a_hi = nullptr, *a_lo = nullptr, *b_hi = nullptr, *b_lo = nullptr;
    float* partial = nullptr;
    std::vector<void*> owned;
    void destroy() { for (void* p : owned) (void)hipFree(p); owned.clear(); }
    bool init(int B_, int H_, int W_, int maxCin_, int maxCout_, bool narrow_layers = false) {
        B = B_; H = H_; W = W_; maxCin = (maxCin_ + 127) / 128 * 128; maxCout = maxCout_;
        P = ((W + 7) & ~7) + 8;
        const size_t K = ((size_t)B * (H + 1) + 1) * P;
        narrow = narrow_layers;
        Kpad = (K + 63) / 64 * 64;                                // whole slices of 64; run() cuts it into splits
        guard = ((size_t)P + 72 + 63) / 64 * 64;                  // row shift (P) + one slice of read-ahead
        ld = guard + Kpad + guard;
        auto zalloc = [&](size_t bytes) -> void* {
            void* p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
            if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return nullptr; }
            owned.push_back(p);
            return p;
        };
        a_hi = (unsigned short*)zalloc((size_t)maxCin * ld * 2); a_lo = (unsigned short*)zalloc((size_t)maxCin * ld * 2);
        b_hi = (unsigned short*)zalloc((size_t)maxCout * ld * 2); b_lo = (unsigned short*)zalloc((size_t)maxCout * ld * 2);
        // room for the split-k partials: 8 splits of the widest layer, and at least 128 MiB so that the narrow layers can be cut finely
        partial_floats = std::max<size_t>((size_t)8 * 9 * maxCin * maxCout, (size_t)32 << 20);
        partial = (float*)zalloc(partial_floats * sizeof(float));
        return a_hi && a_lo && b_hi && b_lo && partial;
    }
    int rewrite(const float* in, int C, int row0, bool gside, int mode, const float* film, int film_bs, const float* affB, float* colpart, bool plain,
                hipStream_t st, int src_mode = 0, const float* affE = nullptr) const {
        if (row0 + C > (gside ? maxCout : maxCin) || C % 4) { hd_set_error("wgrad rewrite: the operand image has too few rows for this tensor"); return -1; }
        hd_prof_begin("wg_prep_kernel", 0.0, (double)B * H * W * C * (4.0 + (plain ? 2.0 : 4.0)), st);   // fp32 in, bf16 hi (+ lo) out
        const dim3 grid(B * H, (C + 255) / 256);
        if (plain)
            hipLaunchKernelGGL(wg_prep_kernel<true>, grid, dim3(256), 0, st, in, C, H, W, P, ld, guard, gside ? b_hi : a_hi, gside ? b_lo : a_lo, mode, film,
                               film_bs, colpart, row0, affB, src_mode, affE);
        else
            hipLaunchKernelGGL(wg_prep_kernel<false>, grid, dim3(256), 0, st, in, C, H, W, P, ld, guard, gside ? b_hi : a_hi, gside ? b_lo : a_lo, mode, film,
                               film_bs, colpart, row0, affB, src_mode, affE);
        conv_prof_end(st);
        return check_launch("wg_prep");
    }
    // splits of a call: the three row taps are separate workgroups, and k is cut until ~one round of workgroups exists (two fit a CU),
    // each keeping at least 8 slices of 64 and the partials fitting their buffer.  (Round 1 cut k only 40 / 8 / 4 ways on the 32x32 /
    // 16x16 / 8x8 maps: 64-128 workgroups on 256 CUs, each looping over the three row taps.)  Measured (MI355X, 64 tiles): the UNet's
    // layers (1-12 tiles per split) are fastest with ONE round of 512 workgroups (39.0 ms per step; 40.7-40.8 at 256 / 768 / 1024),
    // hicedrn's 24-tile layers with 960 (0.96 ms per GEMM; 1.02 at 384).
    int pick_splits(int tiles, int total_slices, size_t per_split_floats, int* slices_per_split, int want = 0) const {
        static const int target_env = getenv("HICDIFF_WG_TARGET") ? atoi(getenv("HICDIFF_WG_TARGET")) : 0;
        const int target = want ? want : target_env ? target_env : (tiles >= 16 ? 1024 : 512);
        int eff = std::max(1, target / tiles);
        eff = std::min<int>(eff, std::max(1, total_slices / 8));
        eff = (int)std::min<size_t>((size_t)eff, partial_floats / (per_split_floats + 1024));   // + room for a bias row per split
        if (eff >= 8) eff &= ~7;
        eff = std::max(eff, 1);
        const int sps = (total_slices + eff - 1) / eff;
        *slices_per_split = sps;
        return (total_slices + sps - 1) / sps;                 // no empty split
    }
    // weight (+ bias) gradient of a KT x KT stride-1 same-padded convolution straight from the NHWC tensors (wgrad_direct_kernel):
    // activations (x0 | x1) [B][H][W][C0 + C1], output gradient g [B][H][W][Cout]; dW in the torch layout, db (optional) [Cout]
    // affA / affB (/ affE), all [B][C0 + C1] or null: the input is silu(x affA + affB) (+ affE) per (sample, channel)
    int run_direct(const float* x0, int C0, const float* x1, int C1, const float* g, int Cout, int KT, float* dW, float* db, hipStream_t st,
                   const float* affA = nullptr, const float* affB = nullptr, const float* affE = nullptr, int src_mode = 0, float scale = 1.f,
                   bool accumulate = false, int aff_bs = 0, float aff_addA = 0.f, bool plain = false) const {
        static const int xcd_group = getenv("HICDIFF_WG_NOXCD") ? 0 : 1;
        const int Cin = C0 + C1;
        if (direct9_ok(Cin, W, KT, src_mode)) return run_direct9(x0, C0, x1, C1, g, Cout, dW, db, st, affA, affB, affE, scale, accumulate, aff_bs, aff_addA, plain);
        if (plain) { hd_set_error("wgrad (direct): the plain-bf16 form exists in the nine-tap kernel only"); return -1; }
        const bool m64 = Cin <= 64;
        const bool aff = affA || affB;
        const int BMh = m64 ? 64 : 128, Mt = (Cin + BMh - 1) / BMh, Nt = Cout / 64, Mpad = Mt * BMh;
        if (Cout % 64 || Cout > 1024 || C0 % 4 || C1 % 4 || (KT != 1 && KT != 3) || !x0 || (C1 && !x1)) { hd_set_error("wgrad (direct): unsupported shape"); return -1; }
        const long long Ktot = (long long)B * (H + 1) * (W + 1);
        if (Ktot >= (1ll << 31) / (W + 2)) { hd_set_error("wgrad (direct): too many pixels for the 32-bit position arithmetic"); return -1; }
        const int ndy = KT == 3 ? 3 : 1, total = (int)((Ktot + 63) / 64);
        const size_t per_split_floats = (size_t)KT * KT * Mpad * Cout;
        if (per_split_floats + 1024 > partial_floats) { hd_set_error("wgrad (direct): the partial buffer is too small for this layer"); return -1; }
        int sps = 0;
        const int eff = pick_splits(Mt * Nt * ndy, total, per_split_floats, &sps);
        WgdArgs a{};
        a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.g = g; a.F = Cout; a.B = B; a.H = H; a.W = W;
        a.magicW = (unsigned)((1ull << 32) / (unsigned)(W + 1)) + 1u; a.magicH = (unsigned)((1ull << 32) / (unsigned)(H + 1)) + 1u;
        a.slices_per_split = sps; a.total_slices = total; a.nsplit = eff; a.Mtiles = Mt; a.Ntiles = Nt; a.M = Mpad;
        a.partial = partial; a.biaspart = db ? partial + (size_t)eff * per_split_floats : nullptr; a.xcd_group = xcd_group;
        a.affA = affA; a.affB = affB; a.affE = affE; a.aff_bs = aff_bs > 0 ? aff_bs : Cin; a.aff_addA = aff_addA; a.src_mode = src_mode;
        if (src_mode && (C1 || aff || m64 || src_mode > 2 || (src_mode == 1 && (KT != 3 || ((H | W) & 1))) || (src_mode == 2 && KT != 1))) {
            hd_set_error("wgrad (direct): unsupported source addressing");
            return -1;
        }
        if (aff && (!affB || KT != 3)) { hd_set_error("wgrad (direct): the affine-input form is the 3 x 3 one and needs the shift array"); return -1; }
        const char* name = KT == 1 ? "wgrad_direct_kernel<true>" : "wgrad_direct_kernel<false>";
        hd_prof_begin(name, 2.0 * KT * KT * Cin * Cout * (double)B * H * W, 4.0 * (Cin + Cout) * (double)B * H * W + 4.0 * eff * KT * KT * Cin * Cout, st);
        const dim3 grid(Mt * Nt * eff, ndy);
        if (src_mode == 1) hipLaunchKernelGGL((wgrad_direct_kernel<false, 2, 0, 1>), grid, dim3(256), 0, st, a);
        else if (src_mode == 2) hipLaunchKernelGGL((wgrad_direct_kernel<true, 2, 0, 2>), grid, dim3(256), 0, st, a);
        else if (KT == 1) { if (m64) hipLaunchKernelGGL((wgrad_direct_kernel<true, 1>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((wgrad_direct_kernel<true, 2>), grid, dim3(256), 0, st, a); }
        else if (aff && affE) { if (m64) hipLaunchKernelGGL((wgrad_direct_kernel<false, 1, 2>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((wgrad_direct_kernel<false, 2, 2>), grid, dim3(256), 0, st, a); }
        else if (aff) { if (m64) hipLaunchKernelGGL((wgrad_direct_kernel<false, 1, 1>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((wgrad_direct_kernel<false, 2, 1>), grid, dim3(256), 0, st, a); }
        else { if (m64) hipLaunchKernelGGL((wgrad_direct_kernel<false, 1>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((wgrad_direct_kernel<false, 2>), grid, dim3(256), 0, st, a); }
        conv_prof_end(st);
        if (check_launch("wgrad direct")) return -3;
        const int nmain = (int)((per_split_floats + 255) / 256);
        hipLaunchKernelGGL(wg_reduce_kernel, dim3((unsigned)(nmain + (db ? (Cout + 255) / 256 : 0))), dim3(256), 0, st, partial, eff, Cin, Mpad, Cout, KT, scale,
                           accumulate ? 1 : 0, dW, a.biaspart, db, nmain);
        return check_launch("wg_reduce");
    }
    // the nine-tap form (wgrad_direct9_kernel: one 8-wave workgroup per CU, activations loaded once): 3 x 3, more than 64 input channels
    static bool direct9_ok(int Cin, int W, int KT, int src_mode) {
        static const bool off = getenv("HICDIFF_WG_NO9") != nullptr;
        return !off && KT == 3 && !src_mode && Cin > 64 && 2 * (W + 1) <= 192;
    }
    int run_direct9(const float* x0, int C0, const float* x1, int C1, const float* g, int Cout, float* dW, float* db, hipStream_t st, const float* affA = nullptr,
                    const float* affB = nullptr, const float* affE = nullptr, float scale = 1.f, bool accumulate = false, int aff_bs = 0, float aff_addA = 0.f,
                    bool plain = false) const {
        static const int xcd_group = getenv("HICDIFF_WG_NOXCD") ? 0 : 1;
        static const int want = getenv("HICDIFF_WG9_TARGET") ? atoi(getenv("HICDIFF_WG9_TARGET")) : 256;
        const int Cin = C0 + C1, KT = 3;
        const int Mt = (Cin + 127) / 128, Nt = Cout / 64, Mpad = Mt * 128;
        const bool aff = affA || affB;
        if (Cout % 64 || Cout > 1024 || C0 % 4 || C1 % 4 || !x0 || (C1 && !x1) || (aff && !affB)) { hd_set_error("wgrad (direct9): unsupported shape"); return -1; }
        const long long Ktot = (long long)B * (H + 1) * (W + 1);
        if (Ktot >= (1ll << 31) / (W + 2)) { hd_set_error("wgrad (direct9): too many pixels for the 32-bit position arithmetic"); return -1; }
        const int total = (int)((Ktot + 63) / 64);
        const size_t per_split_floats = (size_t)KT * KT * Mpad * Cout;
        if (per_split_floats + 1024 > partial_floats) { hd_set_error("wgrad (direct9): the partial buffer is too small for this layer"); return -1; }
        int sps = 0;
        const int eff = pick_splits(Mt * Nt, total, per_split_floats, &sps, want);
        const int nlead = (2 * (W + 1) + 63) / 64, R = nlead == 1 ? 128 : 256;
        WgdArgs a{};
        a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.g = g; a.F = Cout; a.B = B; a.H = H; a.W = W;
        a.magicW = (unsigned)((1ull << 32) / (unsigned)(W + 1)) + 1u; a.magicH = (unsigned)((1ull << 32) / (unsigned)(H + 1)) + 1u;
        a.slices_per_split = sps; a.total_slices = total; a.nsplit = eff; a.Mtiles = Mt; a.Ntiles = Nt; a.M = Mpad;
        a.partial = partial; a.biaspart = db ? partial + (size_t)eff * per_split_floats : nullptr; a.xcd_group = xcd_group;
        a.affA = affA; a.affB = affB; a.affE = affE; a.aff_bs = aff_bs > 0 ? aff_bs : Cin; a.aff_addA = aff_addA; a.src_mode = 0;
        const size_t lds = (size_t)R * 256 * 2 + (size_t)66 * 192 * 2;
        hd_prof_begin(plain ? "wgrad_direct9_kernel<plain bf16>" : "wgrad_direct9_kernel", 2.0 * KT * KT * Cin * Cout * (double)B * H * W, 4.0 * (Cin + Cout) * (double)B * H * W + 4.0 * eff * KT * KT * Cin * Cout, st);
        const dim3 grid(Mt * Nt * eff);
#define HD_W9(AFF_, PL_)                                                                                                     \
        do {                                                                                                                 \
            static bool attr = false;                                                                                        \
            if (!attr) { (void)hipFuncSetAttribute((const void*)wgrad_direct9_kernel<AFF_, PL_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
            hipLaunchKernelGGL((wgrad_direct9_kernel<AFF_, PL_>), grid, dim3(512), lds, st, a, R);                           \
        } while (0)
        if (plain) { if (aff && affE) HD_W9(2, true); else if (aff) HD_W9(1, true); else HD_W9(0, true); }
        else if (aff && affE) HD_W9(2, false); else if (aff) HD_W9(1, false); else HD_W9(0, false);
#undef HD_W9
        conv_prof_end(st);
        if (check_launch("wgrad direct9")) return -3;
        const int nmain = (int)((per_split_floats + 255) / 256);
        hipLaunchKernelGGL(wg_reduce_kernel, dim3((unsigned)(nmain + (db ? (Cout + 255) / 256 : 0))), dim3(256), 0, st, partial, eff, Cin, Mpad, Cout, KT, scale,
                           accumulate ? 1 : 0, dW, a.biaspart, db, nmain);
        return check_launch("wg_reduce");
    }
    // dW[Cout][Cin][KT][KT] (+)= scale * sum over pixels (activation image rows 0..Cin-1, gradient image rows 0..Cout-1)
    int run(int Cin, int Cout, int KT, float scale, bool accumulate, float* dW, bool plain, hipStream_t st) const {
        static const int xcd_group = getenv("HICDIFF_WG_NOXCD") ? 0 : 1;
        // layers with at most 64 input channels take the 64-row tile (no padded half); everything else 128 rows per workgroup
        static const bool no64 = getenv("HICDIFF_WG_NO64") != nullptr;
        const bool m64 = Cin <= 64 && !no64;
        const int BMh = m64 ? 64 : 128, Mt = (Cin + BMh - 1) / BMh, Nt = Cout / 64, Mpad = Mt * BMh;
        if (Cout % 64 || Mpad > maxCin || Cout > maxCout || (KT != 1 && KT != 3)) { hd_set_error("wgrad: unsupported shape"); return -1; }
        const int ndy = KT == 3 ? 3 : 1, total = (int)(Kpad / 64);
        const size_t per_split_floats = (size_t)KT * KT * Mpad * Cout;
        int sps = 0;
        const int eff = pick_splits(Mt * Nt * ndy, total, per_split_floats, &sps);
        // algorithmic figures: KT*KT taps x Cin x Cout outputs over the B*H*W real pixels (3 MFMA flops per product are the kernel's business);
        // bytes: both operand images once (hi + lo) + the partials
        const char* name = KT == 1 ? (plain ? "wgrad_gemm_kernel<true, true>" : "wgrad_gemm_kernel<false, true>")
                                   : (plain ? "wgrad_gemm_kernel<true>" : "wgrad_gemm_kernel<false>");
        hd_prof_begin(name, 2.0 * KT * KT * Cin * Cout * (double)B * H * W, (plain ? 1.0 : 2.0) * 2 * (Cin + Cout) * (double)Kpad + 4.0 * eff * KT * KT * Cin * Cout, st);
        const dim3 grid(Mt * Nt * eff, ndy);
#define HD_WG_LAUNCH(PLAIN_, ONE_)                                                                                          \
        if (m64) hipLaunchKernelGGL((wgrad_gemm_kernel<PLAIN_, ONE_, 1>), grid, dim3(256), 0, st, a_hi, a_lo, b_hi, b_lo, ld, guard, P, sps, total, eff, Mt, Nt, Mpad, Cout, \
                                    partial, xcd_group);                                                                    \
        else hipLaunchKernelGGL((wgrad_gemm_kernel<PLAIN_, ONE_, 2>), grid, dim3(256), 0, st, a_hi, a_lo, b_hi, b_lo, ld, guard, P, sps, total, eff, Mt, Nt, Mpad, Cout, \
                                partial, xcd_group)
        if (KT == 1) { if (plain) HD_WG_LAUNCH(true, true); else HD_WG_LAUNCH(false, true); }
        else { if (plain) HD_WG_LAUNCH(true, false); else HD_WG_LAUNCH(false, false); }
#undef HD_WG_LAUNCH
        conv_prof_end(st);
        if (check_launch("wgrad gemm")) return -3;
        const size_t per = (size_t)KT * Mpad * KT * Cout;
        hipLaunchKernelGGL(wg_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, partial, eff, Cin, Mpad, Cout, KT, scale, accumulate ? 1 : 0, dW);
        return check_launch("wg_reduce");
    }
};

// =====================================================================================================================
struct UnetTrainer;
struct hd_trainer {
    hd_arch_desc arch{};
    UnetTrainer* unet = nullptr;          // set for the UNet (train_unet.inc); the fields below the layout are then unused
    int device = 0, B = 0, S = 0, F = 256, nres = 0, cin0 = 1, tdim = 1024, FW = 512;   // FW: FiLM row width (2F; F for SR3's additive form)
    std::string err;
    struct Slot { std::string name; size_t off, n; int ndim; long long shape[4]; };
    std::vector<Slot> slots;
    size_t nparams = 0;
    // gradient stages (hd_train_stage_*): slot -> stage in completion order, one event per stage recorded by every step
    std::vector<int> slot_stage;
    std::vector<hipEvent_t> stage_ev;
    std::vector<int> block_stage_end;     // hicedrn: stage whose last block is i (recorded after block i's backward), or -1
    int slot_stage_of_block(int i) const { return (nres - 1 - i) * std::min(4, nres) / nres; }
    float* film_part = nullptr;           // [B][tiles per sample][F][2]: the data-gradient convolution's FiLM sums (EP_FILM_SILU_BWD)
    float *temb_act = nullptr, *temb_actT = nullptr; int Bp = 0;      // act(temb) and its transpose ([tdim][Bp], Bp = B rounded up to 64): the FiLM projections' input
    void* fjobs_dev = nullptr; int fjobs_n = 0, ftiles = 0;    // hicedrn: the job table of prep_filters_kernel (every block's two packed images in one launch)
    int objective = 0;                    // hd_train_set_objective: what the network's output is compared with (0 noise, 1 x_start, 2 v)
    const float* loss_w = nullptr;        // hd_train_set_loss_weights: per-sample weights of the next steps' loss (device, [B]); null: 1
    float* stage_snap = nullptr;          // tests: hd_debug_train_stage_snapshot
    const float* cur_grads = nullptr;     // the gradient buffer of the step being queued
    size_t o_head_w = 0, o_head_b = 0, o_t1w = 0, o_t1b = 0, o_t3w = 0, o_t3b = 0, o_bt_w = 0, o_bt_b = 0, o_tail_w = 0, o_tail_b = 0;
    std::vector<size_t> o_mlp_w, o_mlp_b, o_conv_w, o_conv_b;
    // device memory
    std::vector<void*> owned;
    std::vector<ConvW> fwd, bwd;          // body convs + body_tail (index nres)
    ConvW tail_fwd;
    float *wt_tmp = nullptr, *tail_flip = nullptr, *zero_bias = nullptr;
    // weight-gradient geometry and operands
    int plain = 0;                        // bf16 products without the two correction terms (hd_train_set_precision)
    Wgrad wg;                             // operand images + partials of the 256 -> 256 weight gradients
    float *colpart = nullptr, *ctmp = nullptr;
    // activations
    std::vector<float*> X, U, Aact;       // X[0..nres], U[0..nres-1]; Aact[i] = silu(film(U[i])) (288 GB of HBM: keeping it is cheaper than forming it twice more)
    float *Y = nullptr, *out = nullptr, *xt = nullptr, *dout = nullptr, *g0 = nullptr, *g1 = nullptr, *g2 = nullptr, *per = nullptr;
    float *emb = nullptr, *h1pre = nullptr, *temb = nullptr, *film = nullptr, *dfilm = nullptr, *dst = nullptr, *dh1 = nullptr, *fpart = nullptr,
          *spart = nullptr, *mpart = nullptr;
};

static thread_local std::string t_err;
static int tfail(hd_trainer* t, int code, const std::string& msg) { t_err = msg; if (t) t->err = msg; hd_set_error(msg); return code; }

#define TR_TRY(expr)                                                        \
    do {                                                                    \
        int rc_ = (expr);                                                   \
        if (rc_ != 0) return tfail(tr, HD_EHIP, "training step: a launch failed (see hd_last_error)"); \
    } while (0)

static void add_slot(hd_trainer* t, const std::string& name, std::initializer_list<long long> shape, size_t* off) {
    hd_trainer::Slot s; s.name = name; s.off = t->nparams; s.ndim = (int)shape.size(); s.n = 1;
    int i = 0; for (long long d : shape) { s.shape[i++] = d; s.n *= (size_t)d; }
    for (; i < 4; ++i) s.shape[i] = 1;
    if (off) *off = s.off;
    t->nparams += (s.n + 3) & ~(size_t)3;                      // keep every tensor 16-byte aligned in the flat buffer
    t->slots.push_back(s);
}

template <class T> static T* dev_alloc(hd_trainer* t, size_t n, bool zero = false) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
    if (zero && hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { (void)hipFree(p); return nullptr; }
    t->owned.push_back(p);
    return (T*)p;
}

// gradient stages: one event per stage, created once the slot -> stage map exists
static bool stage_events(hd_trainer* t) {
    int n = 0;
    for (int s : t->slot_stage) n = std::max(n, s + 1);
    t->stage_ev.assign(n, nullptr);
    for (auto& e : t->stage_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
    return true;
}
// Stage k's gradients are all in their slots at this point of the stream.  With a snapshot buffer set (hd_debug_train_stage_snapshot: tests) the
// stage's slots are also copied there, in stream order right behind the event, so "nothing writes a slot after its stage's event" can be
// checked without racing: the snapshot must equal the gradients the step ends with.
static void stage_done(hd_trainer* t, int k, hipStream_t st) {
    if (k < 0 || k >= (int)t->stage_ev.size()) return;
    (void)hipEventRecord(t->stage_ev[k], st);
    if (t->stage_snap && t->cur_grads)
        for (size_t j = 0; j < t->slots.size(); ++j)
            if (t->slot_stage[j] == k)
                (void)hipMemcpyAsync(t->stage_snap + t->slots[j].off, t->cur_grads + t->slots[j].off, t->slots[j].n * sizeof(float), hipMemcpyDeviceToDevice, st);
}

#include "train_unet.inc"

extern "C" {

const char* hd_train_last_error(const hd_trainer* t) { return t ? t->err.c_str() : t_err.c_str(); }


void hd_train_destroy(hd_trainer* t) {
    if (!t) return;
    (void)hipDeviceSynchronize();         // nothing of this trainer may still be running when its buffers go
    for (hipEvent_t e : t->stage_ev) if (e) (void)hipEventDestroy(e);
    for (void* p : t->owned) (void)hipFree(p);
    t->wg.destroy();
    if (t->unet) { t->unet->destroy(); delete t->unet; }
    delete t;
}

int hd_train_create(hd_trainer** out, int device, const hd_arch_desc* a, int B, int S) {
    if (!out || !a) return HD_EINVAL;
    *out = nullptr;
    if (a->kind == HD_ARCH_UNET) {
        if (a->dim % 64 || a->n_mults < 1 || a->n_mults > 4 || B < 1 || S < 8 || S > 64 || (S >> (a->n_mults - 1)) < 4 || S % (1 << (a->n_mults - 1)))
            return tfail(nullptr, HD_EINVAL, "UNet training: dim a multiple of 64, at most 4 levels, 8 <= S <= 64 divisible by 2^(levels-1)");
        if (hipSetDevice(device) != hipSuccess) return tfail(nullptr, HD_EHIP, "hipSetDevice failed");
        hd_trainer* t = new hd_trainer();
        t->arch = *a; t->device = device; t->B = B; t->S = S;
        t->unet = new UnetTrainer();
        if (!t->unet->init(t, B, S) || !stage_events(t)) { hd_train_destroy(t); return tfail(nullptr, HD_ENOMEM, "hipMalloc failed while sizing the UNet trainer"); }
        *out = t;
        return HD_OK;
    }
    if (a->kind != HD_ARCH_HICEDRN) return tfail(nullptr, HD_EINVAL, "unknown architecture");
    if (a->dim != 256 || a->number_resnet < 1) return tfail(nullptr, HD_EINVAL, "hicedrn: n_feat must be 256");
    if (B < 1 || S < 8 || S > 64 || S % 8) return tfail(nullptr, HD_EINVAL, "training tiles: 8 <= S <= 64, S a multiple of 8");
    if (hipSetDevice(device) != hipSuccess) return tfail(nullptr, HD_EHIP, "hipSetDevice failed");
    hd_trainer* t = new hd_trainer();
    t->arch = *a; t->device = device; t->B = B; t->S = S; t->F = a->dim; t->nres = a->number_resnet; t->cin0 = a->self_condition ? 2 : 1;
    t->tdim = 4 * t->F;
    t->FW = a->sr3 ? t->F : 2 * t->F;
    const int F = t->F, n = t->nres;
    // flat parameter layout = the reference's state_dict order (tests/golden/param_inventory.json)
    add_slot(t, "head.weight", {F, t->cin0, 3, 3}, &t->o_head_w);
    add_slot(t, "head.bias", {F}, &t->o_head_b);
    add_slot(t, "time_mlp.1.weight", {t->tdim, F}, &t->o_t1w);
    add_slot(t, "time_mlp.1.bias", {t->tdim}, &t->o_t1b);
    add_slot(t, "time_mlp.3.weight", {t->tdim, t->tdim}, &t->o_t3w);
    add_slot(t, "time_mlp.3.bias", {t->tdim}, &t->o_t3b);
    t->o_mlp_w.resize(n); t->o_mlp_b.resize(n); t->o_conv_w.resize(n + 1); t->o_conv_b.resize(n + 1);
    for (int i = 0; i < n; ++i) {
        const std::string p = "body." + std::to_string(i);
        const std::string film = a->sr3 ? ".noise_func.noise_func.0" : ".mlp.1";          // src/model/hicedrn_sr3_Diff.py FeatureWiseAffine
        add_slot(t, p + film + ".weight", {t->FW, t->tdim}, &t->o_mlp_w[i]);
        add_slot(t, p + film + ".bias", {t->FW}, &t->o_mlp_b[i]);
        add_slot(t, p + ".conv.proj.weight", {F, F, 3, 3}, &t->o_conv_w[i]);
        add_slot(t, p + ".conv.proj.bias", {F}, &t->o_conv_b[i]);
    }
    add_slot(t, "body_tail.weight", {F, F, 3, 3}, &t->o_conv_w[n]);
    add_slot(t, "body_tail.bias", {F}, &t->o_conv_b[n]);
    add_slot(t, "tail.weight", {1, F, 3, 3}, &t->o_tail_w);
    add_slot(t, "tail.bias", {1}, &t->o_tail_b);
    t->o_bt_w = t->o_conv_w[n]; t->o_bt_b = t->o_conv_b[n];
    {   // Gradient stages (hd_train_stage_*).  The walk back finishes body_tail / tail first, then the blocks n-1 .. 0 (a block's conv weight is
        // final once both uses of the shared convolution have added their gradient; the FiLM projections of a stage's blocks are taken in one
        // launch when the stage's last block is done), and at the very end the time MLP and the head: four stages of blocks (stage 0 also
        // holds body_tail and tail) + one small final stage.
        const int NS = std::min(4, n);
        auto stage_of = [&](int i) { return t->slot_stage_of_block(i); };            // blocks n-1 .. 0 spread evenly over stages 0 .. NS-1, none empty
        t->slot_stage.assign(t->slots.size(), NS);
        t->block_stage_end.assign(n, -1);
        for (int i = 0; i < n; ++i) {
            const int k = stage_of(i);
            for (size_t j = 0; j < t->slots.size(); ++j)
                if (t->slots[j].off == t->o_conv_w[i] || t->slots[j].off == t->o_conv_b[i] || t->slots[j].off == t->o_mlp_w[i] || t->slots[j].off == t->o_mlp_b[i])
                    t->slot_stage[j] = k;                                             // block i: the shared convolution and the FiLM projection
            if (i == 0 || stage_of(i - 1) != k) t->block_stage_end[i] = k;             // the next block walked (i - 1) belongs to a later stage
        }
        for (size_t j = 0; j < t->slots.size(); ++j)
            if (t->slots[j].off >= t->o_conv_w[n]) t->slot_stage[j] = 0;                  // body_tail, tail: the last four slots
    }

    bool ok = true;
    auto need = [&](void* p) { if (!p) ok = false; return p; };
    const size_t wn = (size_t)9 * F * F;
    t->fwd.resize(n + 1); t->bwd.resize(n + 1);
    for (int i = 0; i <= n && ok; ++i)
        for (ConvW* w : {&t->fwd[i], &t->bwd[i]}) {
            w->KH = w->KW = 3; w->Cin = F; w->Cout = F; w->CoutPad = F; w->ck = 32;
            w->w = (float*)need(dev_alloc<float>(t, wn));
            w->wsplit = (unsigned short*)need(dev_alloc<unsigned short>(t, wn * 2));
        }
    if (ok) {
        // one table for prep_filters_kernel (train_unet.inc): forward image and flipped-transposed data-gradient image of all n + 1 shared
        // 256 -> 256 convolutions in ONE launch per step (round 2: pack, split, flip, pack, split and two memsets per layer: 165 launches + 66
        // memsets, 1.9 ms of a 157 ms step)
        std::vector<FilterJob> jobs(n + 1);
        int tiles = 0;
        for (int i = 0; i <= n; ++i) {
            FilterJob& j = jobs[i];
            j = FilterJob{};
            j.src = (long long)t->o_conv_w[i]; j.cout = F; j.cin_src = F; j.kk = 9; j.standardize = 0; j.unshuffle = 0;
            j.cinF = F; j.tapsF = 9; j.ckF = 32; j.coutpadF = F; j.ckB = 32; j.coutpadB = F;
            j.fw = t->fwd[i].w; j.fws = t->fwd[i].wsplit; j.bw = t->bwd[i].w; j.bws = t->bwd[i].wsplit;
            j.first = i * F; j.tile0 = tiles; tiles += (F / 32) * (F / 32);
        }
        t->fjobs_dev = need(dev_alloc<FilterJob>(t, jobs.size()));
        if (t->fjobs_dev && hipMemcpy(t->fjobs_dev, jobs.data(), jobs.size() * sizeof(FilterJob), hipMemcpyHostToDevice) != hipSuccess) ok = false;
        t->fjobs_n = n + 1; t->ftiles = tiles;
    }
    t->tail_fwd.KH = t->tail_fwd.KW = 3; t->tail_fwd.Cin = F; t->tail_fwd.Cout = 1; t->tail_fwd.CoutPad = 64; t->tail_fwd.ck = 32;
    t->tail_fwd.w = (float*)need(dev_alloc<float>(t, (size_t)9 * F * 64));
    t->tail_fwd.wsplit = (unsigned short*)need(dev_alloc<unsigned short>(t, (size_t)9 * F * 64 * 2));
    t->wt_tmp = (float*)need(dev_alloc<float>(t, wn));
    t->tail_flip = (float*)need(dev_alloc<float>(t, (size_t)9 * F));
    t->zero_bias = (float*)need(dev_alloc<float>(t, F, true));
    if (!t->wg.init(B, S, S, F, F)) ok = false;
    t->colpart = (float*)need(dev_alloc<float>(t, (size_t)B * S * F));
    t->ctmp = (float*)need(dev_alloc<float>(t, (size_t)B * S * F));
    const size_t act = (size_t)B * S * S * F, pix = (size_t)B * S * S;
    t->X.resize(n + 1); t->U.resize(n); t->Aact.resize(n);
    for (int i = 0; i <= n && ok; ++i) t->X[i] = (float*)need(dev_alloc<float>(t, act));
    for (int i = 0; i < n && ok; ++i) t->U[i] = (float*)need(dev_alloc<float>(t, act));
    for (int i = 0; i < n && ok; ++i) t->Aact[i] = (float*)need(dev_alloc<float>(t, act));
    t->Y = (float*)need(dev_alloc<float>(t, act));
    t->g0 = (float*)need(dev_alloc<float>(t, act)); t->g1 = (float*)need(dev_alloc<float>(t, act)); t->g2 = (float*)need(dev_alloc<float>(t, act));
    t->out = (float*)need(dev_alloc<float>(t, pix)); t->xt = (float*)need(dev_alloc<float>(t, pix)); t->dout = (float*)need(dev_alloc<float>(t, pix));
    t->per = (float*)need(dev_alloc<float>(t, 2 * B));
    t->emb = (float*)need(dev_alloc<float>(t, (size_t)B * F)); t->h1pre = (float*)need(dev_alloc<float>(t, (size_t)B * t->tdim));
    t->temb = (float*)need(dev_alloc<float>(t, (size_t)B * t->tdim));
    t->Bp = (B + 63) / 64 * 64;
    t->film_part = (float*)need(dev_alloc<float>(t, (size_t)B * ((S * S + 63) / 64) * F * 2));      // (at most one tile per 64 pixels)
    t->temb_act = (float*)need(dev_alloc<float>(t, (size_t)B * t->tdim)); t->temb_actT = (float*)need(dev_alloc<float>(t, (size_t)t->Bp * t->tdim));
    t->film = (float*)need(dev_alloc<float>(t, (size_t)n * B * 2 * F)); t->dfilm = (float*)need(dev_alloc<float>(t, (size_t)n * B * 2 * F));   // sized for the wider form
    t->dst = (float*)need(dev_alloc<float>(t, (size_t)B * t->tdim)); t->dh1 = (float*)need(dev_alloc<float>(t, (size_t)B * t->tdim));
    const int nchunk = (S * S + 63) / 64;
    t->fpart = (float*)need(dev_alloc<float>(t, (size_t)B * nchunk * 2 * F));
    t->spart = (float*)need(dev_alloc<float>(t, (size_t)B * ((S + 7) / 8) * F * 18));
    t->mpart = (float*)need(dev_alloc<float>(t, (size_t)n * B * t->tdim));
    if (!ok || !stage_events(t)) { hd_train_destroy(t); return tfail(nullptr, HD_ENOMEM, "hipMalloc failed while sizing the trainer (saved activations: 3 per block)"); }
    *out = t;
    return HD_OK;
}

int hd_train_set_precision(hd_trainer* t, int mode) {
    if (!t || (mode != HD_PREC_BF16X3 && mode != 2)) return HD_EINVAL;
    // switching back from plain bf16: the lo images hold stale zeros / old data only where the next rewrite overwrites them
    t->plain = mode == 2 ? 1 : 0;
    return HD_OK;
}

int hd_train_param_count(const hd_trainer* t, long long* total_floats) {
    if (!t) return HD_EINVAL;
    if (total_floats) *total_floats = (long long)t->nparams;
    return (int)t->slots.size();
}

int hd_train_set_objective(hd_trainer* t, int objective) {
    if (!t || objective < 0 || objective > 2) return HD_EINVAL;
    if (objective != 0 && t->arch.sr3) return tfail(t, HD_EINVAL, "the SR3 flavour trains on the noise only (src/hicdiff_sr3.py has no objective)");
    t->objective = objective;
    return HD_OK;
}

int hd_train_set_loss_weights(hd_trainer* t, const float* w) {
    if (!t) return HD_EINVAL;
    if (w && t->arch.sr3) return tfail(t, HD_EINVAL, "the SR3 flavour's loss is a plain mean (src/hicdiff_sr3.py:786-791)");
    t->loss_w = w;
    return HD_OK;
}

int hd_train_stage_count(const hd_trainer* t) { return t ? (int)t->stage_ev.size() : HD_EINVAL; }

int hd_train_slot_stage(const hd_trainer* t, int slot, int* stage) {
    if (!t || !stage || slot < 0 || slot >= (int)t->slot_stage.size()) return HD_EINVAL;
    *stage = t->slot_stage[slot];
    return HD_OK;
}

// `stream` waits (on the device; the host does not block) until the last queued step has written every gradient of `stage`.
int hd_train_stage_wait(hd_trainer* t, int stage, void* stream) {
    if (!t || stage < 0 || stage >= (int)t->stage_ev.size()) return HD_EINVAL;
    if (hipStreamWaitEvent((hipStream_t)stream, t->stage_ev[stage], 0) != hipSuccess) return tfail(t, HD_EHIP, "hipStreamWaitEvent failed");
    return HD_OK;
}

int hd_debug_train_stage_snapshot(hd_trainer* t, float* snapshot) {
    if (!t) return HD_EINVAL;
    t->stage_snap = snapshot;
    return HD_OK;
}

int hd_train_param_slot(const hd_trainer* t, int i, const char** name, long long* offset, long long* shape4, int* ndim) {
    if (!t || i < 0 || i >= (int)t->slots.size()) return HD_EINVAL;
    const auto& s = t->slots[i];
    if (name) *name = s.name.c_str();
    if (offset) *offset = (long long)s.off;
    if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = s.shape[k];
    if (ndim) *ndim = s.ndim;
    return HD_OK;
}

}  // extern "C"

// ---- one training step ---------------------------------------------------------------------------------------------
static int conv3(hd_trainer* tr, const ConvW& w, const float* in, float* out, int ep, float alpha, const float* res, hipStream_t st) {
    ConvArgs a;
    a.in0 = in; a.C0 = tr->F; a.B = tr->B; a.H = tr->S; a.W = tr->S; a.IH = tr->S; a.IW = tr->S; a.stride = 1; a.pad = 1; a.cw = w; a.out = out;
    a.ep = ep; a.alpha = alpha; a.res = res; a.precision = HD_PREC_BF16X3; a.plain_bf16 = tr->plain;
    return launch_conv(a, st, nullptr);
}

static int prep(hd_trainer* tr, const float* in, bool gside, int mode, const float* film, float* colpart, hipStream_t st) {
    return tr->wg.rewrite(in, tr->F, 0, gside, mode, film, tr->FW, nullptr, colpart, tr->plain != 0, st);
}

// dW (+)= scale * wgrad(activation side already in the A image, gradient side already in the G image)
static int wgrad(hd_trainer* tr, float scale, bool accumulate, float* dW, hipStream_t st) {
    return tr->wg.run(tr->F, tr->F, 3, scale, accumulate, dW, tr->plain != 0, st);
}

// db (+)= scale * column sums of the [B*S][F] row partials the last G-side prep wrote (two short serial stages)
static int colsum(hd_trainer* tr, float scale, bool accumulate, float* db, hipStream_t st) {
    hipLaunchKernelGGL(sum_rows_kernel, dim3((tr->F + 255) / 256, tr->B), dim3(256), 0, st, tr->colpart, tr->S, tr->F, 1.f, 0, tr->ctmp);
    hipLaunchKernelGGL(sum_rows_kernel, dim3((tr->F + 255) / 256, 1), dim3(256), 0, st, tr->ctmp, tr->B, tr->F, scale, accumulate ? 1 : 0, db);
    return check_launch("colsum");
}

extern "C" int hd_train_loss_backward(hd_trainer* tr, const float* params, float* grads, const float* x_start, const float* cond, const void* t, int t_kind,
                                      const float* noise, const float* a_t, const float* s_t, int l2, float* loss, void* stream) {
    if (!tr || !params || !grads || !x_start || !t || !noise || !a_t || !s_t || !loss) return HD_EINVAL;
    tr->cur_grads = grads;
    if (tr->unet) {
        if ((tr->arch.self_condition != 0) != (cond != nullptr)) return tfail(tr, HD_EINVAL, "cond must be given iff self_condition");
        if ((t_kind == HD_T_FLOAT32) != (tr->arch.sr3 != 0)) return tfail(tr, HD_EINVAL, "SR3 nets take the continuous noise level (float32); the others integer timesteps");
        tr->err.clear();
        const int rc = tr->unet->step(params, grads, x_start, cond, t, noise, a_t, s_t, l2, loss, (hipStream_t)stream);   // t: int64 steps, or float levels (SR3)
        return rc == 0 ? HD_OK : tfail(tr, rc == -4 ? HD_ENOMEM : HD_EHIP, "UNet training step: " + (tr->unet->why.empty() ? std::string("a launch failed") : tr->unet->why));
    }
    if ((tr->cin0 == 2) != (cond != nullptr)) return tfail(tr, HD_EINVAL, "cond must be given iff self_condition");
    hipStream_t st = (hipStream_t)stream;
    const int F = tr->F, S = tr->S, B = tr->B, n = tr->nres, TD = tr->tdim, HW = S * S, FW = tr->FW;
    const bool sr3 = tr->arch.sr3 != 0;
    const int t_float = t_kind == HD_T_FLOAT32 ? 1 : 0;
    if (t_kind != HD_T_INT64 && t_kind != HD_T_FLOAT32) return tfail(tr, HD_EINVAL, "t_kind must be HD_T_INT64 or HD_T_FLOAT32");
    if (sr3 != (t_float != 0)) return tfail(tr, HD_EINVAL, "SR3 nets take the continuous noise level (float32); the others integer timesteps");
    const size_t act = (size_t)B * HW * F;
    tr->err.clear();
    // ---- weights: forward image and flipped-transposed image for the data gradient of every shared convolution, one launch
    static const bool per_layer_prep = getenv("HICDIFF_TRAIN_PREP_PER_LAYER") != nullptr;       // A/B: round 2's five launches per layer
    if (!per_layer_prep) {
        hipLaunchKernelGGL(prep_filters_kernel, dim3(tr->ftiles), dim3(256), 0, st, params, (const FilterJob*)tr->fjobs_dev, tr->fjobs_n, (const double2*)nullptr);
        TR_TRY(check_launch("prep_filters"));
    }
    for (int i = 0; i <= n; ++i) {
        tr->fwd[i].bias = const_cast<float*>(params + tr->o_conv_b[i]);
        tr->bwd[i].bias = nullptr;
        if (!per_layer_prep) continue;
        const float* w = params + tr->o_conv_w[i];
        TR_TRY(launch_pack_conv(w, tr->fwd[i].w, F, F, 3, 3, F, 0, 0, st));
        TR_TRY(launch_split_conv(tr->fwd[i].w, tr->fwd[i].wsplit, 9, F, F, 16, st));
        hipLaunchKernelGGL(flip_weight_kernel, dim3((unsigned)(((size_t)9 * F * F + 255) / 256)), dim3(256), 0, st, w, F, F, tr->wt_tmp);
        TR_TRY(check_launch("flip_weight"));
        TR_TRY(launch_pack_conv(tr->wt_tmp, tr->bwd[i].w, F, F, 3, 3, F, 0, 0, st));
        TR_TRY(launch_split_conv(tr->bwd[i].w, tr->bwd[i].wsplit, 9, F, F, 16, st));
    }
    TR_TRY(launch_pack_conv(params + tr->o_tail_w, tr->tail_fwd.w, 1, F, 3, 3, 64, 0, 0, st));
    TR_TRY(launch_split_conv(tr->tail_fwd.w, tr->tail_fwd.wsplit, 9, F, 64, 16, st));
    tr->tail_fwd.bias = const_cast<float*>(params + tr->o_tail_b);
    hipLaunchKernelGGL(flip_weight_kernel, dim3((9 * F + 255) / 256), dim3(256), 0, st, params + tr->o_tail_w, 1, F, tr->tail_flip);   // -> [F][1][3][3]
    TR_TRY(check_launch("flip_tail"));

    // ---- forward (src/hicdiff.py:694-700,711-747; src/model/hicedrn_Diff.py:267-289)
    TR_TRY(launch_q_sample(x_start, noise, a_t, s_t, tr->xt, B, S, st));
    hipLaunchKernelGGL(sin_emb_kernel, dim3(B), dim3(256), 0, st, t, t_float, sr3 ? 1 : 0, F, tr->emb);
    TR_TRY(check_launch("sin_emb"));
    hipLaunchKernelGGL(lin_fwd_kernel, dim3((TD + 3) / 4), dim3(256), 0, st, tr->emb, F, params + tr->o_t1w, params + tr->o_t1b, B, F, TD, 0, tr->h1pre, TD, (size_t)0, (size_t)0, (size_t)0);
    hipLaunchKernelGGL(lin_fwd_kernel, dim3((TD + 3) / 4), dim3(256), 0, st, tr->h1pre, TD, params + tr->o_t3w, params + tr->o_t3b, B, TD, TD, 2, tr->temb, TD, (size_t)0, (size_t)0, (size_t)0);
    const size_t lstride = n > 1 ? tr->o_mlp_w[1] - tr->o_mlp_w[0] : 0;        // every block's slots have the same sizes: a constant stride
    // non-SR3: Linear(SiLU(temb)) -> (scale, shift) (src/model/hicedrn_Diff.py:185-199); SR3: Linear(temb) -> shift (hicedrn_sr3_Diff.py:167-183)
    static const bool film_old = getenv("HICDIFF_TRAIN_FILM_LIN_OLD") != nullptr;      // A/B: round 2's one-wave-per-(column, sample) kernels with the activation inside
    const bool film_t = !film_old && TD % 4 == 0;
    if (film_t) {
        hipLaunchKernelGGL(act_rows_kernel, dim3((unsigned)(((size_t)tr->Bp * TD + 255) / 256)), dim3(256), 0, st, tr->temb, B, TD, sr3 ? 0 : 1, tr->Bp, tr->temb_act, tr->temb_actT);
        hipLaunchKernelGGL(lin_fwd_t_kernel, dim3((FW + 15) / 16, n, tr->Bp / 64), dim3(256), 0, st, tr->temb_actT, tr->Bp, params + tr->o_mlp_w[0], params + tr->o_mlp_b[0], B,
                           TD, FW, tr->film, FW, lstride, lstride, (size_t)B * FW);
    } else {
        hipLaunchKernelGGL(lin_fwd_kernel, dim3((FW + 3) / 4, n), dim3(256), 0, st, tr->temb, TD, params + tr->o_mlp_w[0], params + tr->o_mlp_b[0], B, TD, FW, sr3 ? 0 : 1,
                           tr->film, FW, lstride, lstride, (size_t)B * FW);
    }
    TR_TRY(check_launch("time/film forward"));
    static const bool film_pass = getenv("HICDIFF_TRAIN_FILM_PASS") != nullptr;      // A/B: the separate FiLM + SiLU pass of round 2
    const bool film_in_epilogue = !film_pass;
    TR_TRY(launch_conv_small_cin(tr->xt, cond, params + tr->o_head_w, params + tr->o_head_b, tr->X[0], B, S, 3, tr->cin0, F, st));
    const size_t n4 = act / 4;
    for (int i = 0; i < n; ++i) {
        const float* film = tr->film + (size_t)i * B * FW;
        if (film_in_epilogue) {
            // the first convolution's epilogue writes u (kept for the backward pass) AND a = silu(film(u)): the second convolution, and later the
            // weight gradient of its use, read a as it is (a transforming loader cost 14 % of a convolution, a separate pass 2 x 268 MB)
            ConvArgs c1;
            c1.in0 = tr->X[i]; c1.C0 = F; c1.B = B; c1.H = S; c1.W = S; c1.IH = S; c1.IW = S; c1.stride = 1; c1.pad = 1; c1.cw = tr->fwd[i];
            c1.out = tr->Aact[i]; c1.pre_out = tr->U[i]; c1.precision = HD_PREC_BF16X3; c1.plain_bf16 = tr->plain;
            c1.ep = FW > F ? EP_FILM_SILU : EP_ADD_SILU; c1.epScale = film; c1.epShift = film + (FW - F); c1.ep_bstride = FW;
            TR_TRY(launch_conv(c1, st, nullptr));
        } else {
            TR_TRY(conv3(tr, tr->fwd[i], tr->X[i], tr->U[i], 0, 1.f, nullptr, st));
            hipLaunchKernelGGL(film_silu_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, tr->U[i], film, FW, HW, F, n4, tr->Aact[i]);
            TR_TRY(check_launch("film_silu_fwd"));
        }
        TR_TRY(conv3(tr, tr->fwd[i], tr->Aact[i], tr->X[i + 1], EP_RES, 0.1f, tr->X[i], st));
    }
    TR_TRY(conv3(tr, tr->fwd[n], tr->X[n], tr->Y, EP_RES, 1.f, tr->X[0], st));
    {
        ConvArgs o;
        o.in0 = tr->Y; o.C0 = F; o.B = B; o.H = S; o.W = S; o.IH = S; o.IW = S; o.stride = 1; o.pad = 1; o.cw = tr->tail_fwd; o.out = tr->out;
        o.precision = HD_PREC_BF16X3;
        TR_TRY(launch_conv(o, st, nullptr));
    }
    hipLaunchKernelGGL(loss_grad_kernel, dim3(B), dim3(256), 0, st, tr->out, noise, HW, B, l2 ? 1 : 0, tr->per, tr->dout, tr->objective, x_start, a_t, s_t, tr->loss_w);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, st, tr->per, B, loss, grads + tr->o_tail_b);
    TR_TRY(check_launch("loss"));

    // ---- backward
    const int RB = 8, nrb = (S + RB - 1) / RB;
    // tail: dW[0][ci][tap] = sum_p Y[p + tap][ci] dout[p]; db = sum dout; dY = conv(dout, flipped tail weight)
    hipLaunchKernelGGL((small_conv_wgrad_kernel<3, 1>), dim3(nrb, B), dim3(256), (size_t)(RB + 2) * (S + 2) * sizeof(float), st, tr->Y, tr->dout, (const float*)nullptr, 1, S, F, RB,
                       -1, tr->spart);
    hipLaunchKernelGGL(sum_rows_kernel, dim3((F * 9 + 255) / 256, 1), dim3(256), 0, st, tr->spart, B * nrb, F * 9, 1.f, 0, grads + tr->o_tail_w);
    TR_TRY(check_launch("tail wgrad"));
    float* dY = tr->g1;
    TR_TRY(launch_conv_small_cin(tr->dout, nullptr, tr->tail_flip, tr->zero_bias, dY, B, S, 3, 1, F, st));
    // body_tail: dX_n = dgrad(dY); dW = wgrad(X_n, dY); the skip r sends dY to the head output as well
    // weight gradients: straight from the NHWC tensors (wgrad_direct_kernel) in the default arithmetic; operand rewrite + GEMM for plain bf16
    static const bool no_direct = getenv("HICDIFF_WG_NODIRECT") != nullptr;
    const bool plain = tr->plain != 0;
    const bool direct = !no_direct && (!plain || Wgrad::direct9_ok(F, S, 3, 0));     // the plain-bf16 arithmetic has the nine-tap direct kernel only
    if (direct) {
        TR_TRY(tr->wg.run_direct(tr->X[n], F, nullptr, 0, dY, F, 3, grads + tr->o_conv_w[n], grads + tr->o_conv_b[n], st, nullptr, nullptr, nullptr, 0, 1.f, false, 0, 0.f, plain));
    } else {
        TR_TRY(prep(tr, tr->X[n], false, 0, nullptr, nullptr, st));
        TR_TRY(prep(tr, dY, true, 0, nullptr, tr->colpart, st));
        TR_TRY(wgrad(tr, 1.f, false, grads + tr->o_conv_w[n], st));
        TR_TRY(colsum(tr, 1.f, false, grads + tr->o_conv_b[n], st));
    }
    float* dx = tr->g2;                       // gradient w.r.t. the current block output
    float* da = tr->g0;
    TR_TRY(conv3(tr, tr->bwd[n], dY, dx, 0, 1.f, nullptr, st));
    const int nchunk = (HW + 63) / 64;
    for (int i = n - 1; i >= 0; --i) {
        const float* film = tr->film + (size_t)i * B * FW;
        float* dW = grads + tr->o_conv_w[i];
        float* db = grads + tr->o_conv_b[i];
        // second use of the conv: y = 0.1 conv(a) + x
        if (direct) {                                                            // a = silu(u (scale + 1) + shift) (FW == F: shift only)
            TR_TRY(tr->wg.run_direct(tr->Aact[i], F, nullptr, 0, dx, F, 3, dW, db, st, nullptr, nullptr, nullptr, 0, 0.1f, false, 0, 0.f, plain));
        } else {
            TR_TRY(prep(tr, tr->Aact[i], false, 0, nullptr, nullptr, st));           // a = silu(film(u)), kept by the forward pass
            TR_TRY(prep(tr, dx, true, 0, nullptr, tr->colpart, st));
            TR_TRY(wgrad(tr, 0.1f, false, dW, st));
            TR_TRY(colsum(tr, 0.1f, false, db, st));
        }
        // dL/da / 0.1 by the data-gradient convolution, then through a = silu(film(u)): in the convolution's epilogue where its tiling allows the
        // per-sample sums (one sample per tile), else as round 2's separate pass (3 x 268 MB per block at 64 tiles)
        ConvArgs dg;
        dg.in0 = dx; dg.C0 = F; dg.B = B; dg.H = S; dg.W = S; dg.IH = S; dg.IW = S; dg.stride = 1; dg.pad = 1; dg.cw = tr->bwd[i]; dg.out = da;
        dg.precision = HD_PREC_BF16X3; dg.plain_bf16 = tr->plain;
        dg.ep = EP_FILM_SILU_BWD; dg.alpha = 0.1f; dg.res = tr->U[i]; dg.epScale = FW > F ? film : nullptr; dg.epShift = film + (FW - F); dg.ep_bstride = FW;
        dg.gn_part = tr->film_part;
        static const bool film_bwd_pass = getenv("HICDIFF_TRAIN_FILM_BWD_PASS") != nullptr;       // A/B: round 2's separate pass
        const int fslots = film_bwd_pass || !conv_film_bwd_ok(dg) ? 0 : conv_gn_slots(dg);
        if (fslots > 0 && (size_t)fslots <= (size_t)(HW + 63) / 64) {
            TR_TRY(launch_conv(dg, st, nullptr));
            hipLaunchKernelGGL(film_part_reduce_kernel, dim3((F + 255) / 256, B), dim3(256), 0, st, tr->film_part, fslots, F, FW, tr->dfilm + (size_t)i * B * FW);
            TR_TRY(check_launch("film_part_reduce"));
        } else {
            TR_TRY(conv3(tr, tr->bwd[i], dx, da, 0, 1.f, nullptr, st));
            hipLaunchKernelGGL(film_silu_bwd_kernel, dim3(nchunk, B), dim3(256), 0, st, da, tr->U[i], film, FW, HW, F, 64, 0.1f, tr->fpart);
            hipLaunchKernelGGL(sum_rows_kernel, dim3((FW + 255) / 256, B), dim3(256), 0, st, tr->fpart, nchunk, FW, 1.f, 0, tr->dfilm + (size_t)i * B * FW);
            TR_TRY(check_launch("film_silu_bwd"));
        }
        // first use: u = conv(x)
        if (direct) {
            TR_TRY(tr->wg.run_direct(tr->X[i], F, nullptr, 0, da, F, 3, dW, db, st, nullptr, nullptr, nullptr, 0, 1.f, true, 0, 0.f, plain));
        } else {
            TR_TRY(prep(tr, tr->X[i], false, 0, nullptr, nullptr, st));
            TR_TRY(prep(tr, da, true, 0, nullptr, tr->colpart, st));
            TR_TRY(wgrad(tr, 1.f, true, dW, st));
            TR_TRY(colsum(tr, 1.f, true, db, st));
        }
        float* nxt = (dx == tr->g2) ? tr->Y : tr->g2;                            // Y is free once its gradients are taken; g1 keeps dY
        TR_TRY(conv3(tr, tr->bwd[i], da, nxt, EP_RES, 1.f, dx, st));            // dx_i = dgrad(du) + dx_{i+1}
        dx = nxt;
        if (tr->block_stage_end[i] >= 0) {
            // the stage's blocks are i .. hi: their FiLM projections (Linear(SiLU(temb)) per block) in one launch, then the stage's event
            int hi = i;
            while (hi + 1 < n && tr->slot_stage_of_block(hi + 1) == tr->block_stage_end[i]) ++hi;
            hipLaunchKernelGGL(lin_bwd_w_kernel, dim3((TD + 255) / 256, FW, hi - i + 1), dim3(256), 0, st, tr->dfilm + (size_t)i * B * FW, FW,
                               film_t ? tr->temb_act : tr->temb, TD, B, TD, FW, film_t ? 0 : (sr3 ? 0 : 1), grads + tr->o_mlp_w[i], grads + tr->o_mlp_b[i],
                               (size_t)B * FW, lstride, lstride);
            TR_TRY(check_launch("film projection wgrad"));
            stage_done(tr, tr->block_stage_end[i], st);
        }
    }
    // head: d(head output) = dx + dY (the skip r); dW[co][cin][tap] = sum_p in_cin[p + tap] d[p][co]
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, dx, dY, n4, da);
    const float* s0 = tr->cin0 == 2 ? cond : tr->xt;
    hipLaunchKernelGGL((small_conv_wgrad_kernel<3, 1>), dim3(nrb, B), dim3(256), (size_t)tr->cin0 * (RB + 2) * (S + 2) * sizeof(float), st, da, s0, tr->xt, tr->cin0, S, F,
                       RB, 1, tr->spart);
    hipLaunchKernelGGL(sum_rows_kernel, dim3((F * 9 * tr->cin0 + 255) / 256, 1), dim3(256), 0, st, tr->spart, B * nrb, F * 9 * tr->cin0, 1.f, 0, grads + tr->o_head_w);
    hipLaunchKernelGGL(sum_rows_kernel, dim3((F + 255) / 256, B * S), dim3(256), 0, st, da, S, F, 1.f, 0, tr->colpart);
    TR_TRY(check_launch("head wgrad"));
    TR_TRY(colsum(tr, 1.f, false, grads + tr->o_head_b, st));
    // the time MLP (the weight gradients of the FiLM projections were taken stage by stage above)
    hipLaunchKernelGGL(lin_bwd_x_kernel, dim3((TD + 255) / 256, B, n), dim3(256), 0, st, tr->dfilm, FW, params + tr->o_mlp_w[0], B, TD, FW, (const float*)nullptr, 0,
                       0, 0, tr->mpart, TD, (size_t)B * FW, lstride, (size_t)B * TD);
    hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)(((size_t)B * TD + 255) / 256), 1), dim3(256), 0, st, tr->mpart, n, B * TD, 1.f, 0, tr->dst);
    if (!sr3) hipLaunchKernelGGL(dact_mul_kernel, dim3((unsigned)(((size_t)B * TD + 255) / 256)), dim3(256), 0, st, tr->dst, tr->temb, (size_t)B * TD, 1);   // d silu(temb) -> d temb
    hipLaunchKernelGGL(lin_bwd_w_kernel, dim3((TD + 255) / 256, TD, 1), dim3(256), 0, st, tr->dst, TD, tr->h1pre, TD, B, TD, TD, 2, grads + tr->o_t3w, grads + tr->o_t3b,
                       (size_t)0, (size_t)0, (size_t)0);
    hipLaunchKernelGGL(lin_bwd_x_kernel, dim3((TD + 255) / 256, B, 1), dim3(256), 0, st, tr->dst, TD, params + tr->o_t3w, B, TD, TD, tr->h1pre, TD, 2, 0, tr->dh1, TD,
                       (size_t)0, (size_t)0, (size_t)0);
    hipLaunchKernelGGL(lin_bwd_w_kernel, dim3((F + 255) / 256, TD, 1), dim3(256), 0, st, tr->dh1, TD, tr->emb, F, B, F, TD, 0, grads + tr->o_t1w, grads + tr->o_t1b,
                       (size_t)0, (size_t)0, (size_t)0);
    TR_TRY(check_launch("time mlp backward"));
    stage_done(tr, (int)tr->stage_ev.size() - 1, st);           // head, time MLP and every block's FiLM projection: the last stage
    return HD_OK;
}

extern "C" int hd_adam_step(float* params, const float* grads, float* m, float* v, long long n, float lr, float b1, float b2, float eps, int step,
                            float grad_scale, void* stream) {
    if (!params || !grads || !m || !v || n < 0 || step < 1) return HD_EINVAL;
    if (n == 0) return HD_OK;
    const double c1 = 1.0 - std::pow((double)b1, step), c2 = 1.0 - std::pow((double)b2, step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, (size_t)n, b1, b2, eps,
                       (float)(lr / c1), (float)(1.0 / std::sqrt(c2)), grad_scale);
    return check_launch("adam") == 0 ? HD_OK : HD_EHIP;
}

// ---- first / last convolutions of the UNet as components ----------------------------------------------------------------------------------
// dW[C][J][KS][KS] = sum_p in_j[p + tap] * g[p][c] (the first convolution: J = 1 or 2 single-channel input planes, g = gradient of its
// C-channel output); scratch: B * ceil(S/8) * C * J*KS*KS floats
int launch_first_conv_wgrad(const float* g, const float* in0, const float* in1, int J, int B, int S, int C, int KS, float* scratch, float* dW, hipStream_t st) {
    const int RB = 8, nrb = (S + RB - 1) / RB, R = KS / 2, T = KS * KS;
    const size_t lds = (size_t)J * (RB + 2 * R) * (S + 2 * R) * sizeof(float);
    if (KS == 7 && C == 64) hipLaunchKernelGGL((small_conv_wgrad_kernel<7, 4>), dim3(nrb, B), dim3(256), lds, st, g, in0, in1, J, S, C, RB, 1, scratch);
    else if (KS == 7 && C == 128) hipLaunchKernelGGL((small_conv_wgrad_kernel<7, 2>), dim3(nrb, B), dim3(256), lds, st, g, in0, in1, J, S, C, RB, 1, scratch);
    else if (KS == 7) hipLaunchKernelGGL((small_conv_wgrad_kernel<7, 1>), dim3(nrb, B), dim3(256), lds, st, g, in0, in1, J, S, C, RB, 1, scratch);
    else if (KS == 3) hipLaunchKernelGGL((small_conv_wgrad_kernel<3, 1>), dim3(nrb, B), dim3(256), lds, st, g, in0, in1, J, S, C, RB, 1, scratch);
    else { hd_set_error("first-conv weight gradient: 3x3 or 7x7"); return -1; }
    hipLaunchKernelGGL(sum_rows_kernel, dim3((C * J * T + 255) / 256, 1), dim3(256), 0, st, scratch, B * nrb, C * J * T, 1.f, 0, dW);
    return check_launch("first conv wgrad");
}
// final 1x1 convolution C -> 1: dx[p][c] = dout[p] w[c]; dw[c] = sum_p x[p][c] dout[p]; scratch: ceil(P/64) * C floats
int launch_rowdot_bwd(const float* x, const float* dout, const float* w, size_t P, int C, float* scratch, float* dx, float* dw, hipStream_t st) {
    const int rows = 64;
    const unsigned nb = (unsigned)((P + rows - 1) / rows);
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(nb), dim3(256), 0, st, x, dout, w, P, C, rows, dx, scratch);
    hipLaunchKernelGGL(sum_rows_kernel, dim3((C + 255) / 256, 1), dim3(256), 0, st, scratch, (int)nb, C, 1.f, 0, dw);
    return check_launch("rowdot backward");
}
extern "C" int hd_debug_first_conv_wgrad(const float* g, const float* in0, const float* in1, int J, int B, int S, int C, int KS, float* dW, void* stream) {
    if (!g || !in0 || !dW || J < 1 || J > 2 || (J == 2 && !in1)) return HD_EINVAL;
    float* scratch = nullptr;
    if (hipMalloc(&scratch, (size_t)B * ((S + 7) / 8) * C * J * KS * KS * sizeof(float)) != hipSuccess) return HD_ENOMEM;
    const int rc = launch_first_conv_wgrad(g, in0, in1, J, B, S, C, KS, scratch, dW, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(scratch);
    return rc ? (rc == -1 ? HD_EINVAL : HD_EHIP) : HD_OK;
}
extern "C" int hd_debug_rowdot_bwd(const float* x, const float* dout, const float* w, long long P, int C, float* dx, float* dw, void* stream) {
    if (!x || !dout || !w || !dx || !dw || P < 1) return HD_EINVAL;
    float* scratch = nullptr;
    if (hipMalloc(&scratch, ((size_t)(P + 63) / 64) * C * sizeof(float)) != hipSuccess) return HD_ENOMEM;
    const int rc = launch_rowdot_bwd(x, dout, w, (size_t)P, C, scratch, dx, dw, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(scratch);
    return rc ? HD_EHIP : HD_OK;
}

// ---- gradient routing of the resampling layers -------------------------------------------------------------------------------------
namespace {
// Upsample backward: dx[b][y][x][c] = sum of the 2x2 block of the upsampled map's gradient.  g: [B][2H][2W][C], dx: [B][H][W][C].
__global__ __launch_bounds__(256) void sum_pool2_kernel(const float* __restrict__ g, int H, int W, int C4, size_t n4, float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4); size_t r = i / C4;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H), b = (int)(r / H);
    const float4* p = reinterpret_cast<const float4*>(g) + (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C4 + c;
    const float4 a = p[0], bq = p[C4], cq = p[(size_t)2 * W * C4], d = p[(size_t)2 * W * C4 + C4];
    reinterpret_cast<float4*>(dx)[i] = make_float4((a.x + bq.x) + (cq.x + d.x), (a.y + bq.y) + (cq.y + d.y), (a.z + bq.z) + (cq.z + d.z), (a.w + bq.w) + (cq.w + d.w));
}
// Downsample backward: dx[b][2y+p1][2x+p2][c] = g[b][y][x][c*4 + p1*2 + p2].  g: [B][H][W][4C], dx: [B][2H][2W][C].
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const float* __restrict__ g, int H, int W, int C, size_t n, float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // over g's quads: (b, y, x, c)
    if (i >= n) return;
    const int c = (int)(i % C); size_t r = i / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H), b = (int)(r / H);
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    float* d = dx + (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
    d[0] = v.x; d[C] = v.y; d[(size_t)2 * W * C] = v.z; d[(size_t)2 * W * C + C] = v.w;
}
}  // namespace

int launch_sum_pool2(const float* g, int B, int H, int W, int C, float* dx, hipStream_t st) {
    const size_t n4 = (size_t)B * H * W * C / 4;
    hipLaunchKernelGGL(sum_pool2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, g, H, W, C / 4, n4, dx);
    return check_launch("sum_pool2");
}
int launch_pixel_shuffle(const float* g, int B, int H, int W, int C, float* dx, hipStream_t st) {
    const size_t n = (size_t)B * H * W * C;
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, H, W, C, n, dx);
    return check_launch("pixel_shuffle");
}
extern "C" int hd_debug_resample_bwd(const float* g, int B, int H, int W, int C, int which, float* dx, void* stream) {
    if (!g || !dx || B < 1 || C % 4) return HD_EINVAL;
    const int rc = which == 1 ? launch_sum_pool2(g, B, H, W, C, dx, (hipStream_t)stream) : launch_pixel_shuffle(g, B, H, W, C, dx, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    return rc ? HD_EHIP : HD_OK;
}

extern "C" int hd_debug_conv_wgrad_direct(const float* x0, int C0, const float* x1, int C1, const float* g, int B, int H, int W, int Cout, int KT, float* dW,
                                          float* db, const float* affA, const float* affB, const float* affE, int src_mode, void* stream) {
    if (!x0 || !g || !dW || B < 1 || H < 1 || W < 1 || C0 % 4 || C1 % 4 || (C1 && !x1) || Cout % 64 || (KT != 1 && KT != 3)) return HD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    Wgrad wg;
    if (!wg.init(B, H, W, 128, 64)) { wg.destroy(); return HD_ENOMEM; }      // only its partial buffer is used
    const int rc = wg.run_direct(x0, C0, x1, C1, g, Cout, KT, dW, db, st, affA, affB, affE, src_mode);
    (void)hipStreamSynchronize(st);
    wg.destroy();
    return rc ? HD_EHIP : HD_OK;
}

#ifdef HD_STAMPS
extern "C" int hd_debug_wgd_stamps(unsigned long long* out, int nwg) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgd_stamps), sizeof(unsigned long long) * 5 * nwg) == hipSuccess ? 0 : -3;
}
#endif

// ---- test-only entry (include/hicdiff_hip_debug.h): the weight-gradient component on arbitrary shapes ---------------------------
extern "C" int hd_debug_conv_wgrad(const float* x0, int C0, const float* x1, int C1, const float* g, int B, int H, int W, int Cout, int KT, const float* affA,
                                   const float* affB, int plain, float* dW, void* stream) {
    if (!x0 || !g || !dW || B < 1 || H < 1 || W < 1 || W > 64 || C0 % 4 || C1 % 4 || (C1 && !x1) || Cout % 64 || (KT != 1 && KT != 3)) return HD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    Wgrad wg;
    const int Cin = C0 + C1;
    if (!wg.init(B, H, W, Cin, Cout)) { wg.destroy(); return HD_ENOMEM; }
    const int src_mode = plain >> 1;                           // bits above bit 0 of `plain`: 1 = x0 is the half-size map to upsample, 2 = x0 is the double-size map to unshuffle
    plain &= 1;
    int rc = wg.rewrite(x0, C0, 0, false, affA ? 2 : 0, affA, Cin, affB, nullptr, plain != 0, st, src_mode);
    if (!rc && C1) rc = wg.rewrite(x1, C1, C0, false, affA ? 2 : 0, affA ? affA + C0 : nullptr, Cin, affA ? affB + C0 : nullptr, nullptr, plain != 0, st);
    if (!rc) rc = wg.rewrite(g, Cout, 0, true, 0, nullptr, 0, nullptr, nullptr, plain != 0, st);
    if (!rc) rc = wg.run(Cin, Cout, KT, 1.f, false, dW, plain != 0, st);
    (void)hipStreamSynchronize(st);
    wg.destroy();
    return rc ? HD_EHIP : HD_OK;
}
