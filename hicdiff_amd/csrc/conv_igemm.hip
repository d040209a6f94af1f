// Implicit-GEMM convolution for gfx950 on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel serves every "wide" convolution of the two epsilon-networks (reference call sites:
// WeightStandardizedConv2d / Conv2d 3x3 src/hicdiff.py:75,158,320,336; 1x1 qkv / to_out / res_conv
// :183,205,208,236,237; pixel-unshuffle Downsample :78-82 expressed as a 2x2 stride-2 conv; hicedrn's
// 3x3 body src/model/hicedrn_Diff.py:169-208,256-262):
//
//   out[pix][n] = epilogue( bias[n] + sum_{tap,c} T(in)[pix (+) tap][c] * W[tap][c][n] )
//
//   * activations NHWC fp32; GEMM M = output pixels (a TB x TH x TW window per workgroup, so small
//     feature maps fold several images into one 128-row tile), N = Cout, K = taps * Cin;
//   * the input window (with halo) of one CK-channel slice is staged ONCE in LDS and re-used by all
//     taps; T() is applied while staging: GroupNorm-apply + FiLM + SiLU of the producing conv
//     (per-(sample,channel) affine A,B[,E]), channel LayerNorm, nearest x2 upsample (index math),
//     channel concat of two tensors (two base pointers) -- none of these is ever materialised;
//   * weight slabs [CK][BN] are double-buffered in LDS and prefetched into registers one tap ahead;
//   * 4 waves (2 x 2), each TM x TN tiles of 32x32 accumulators; lanes of a 32x32 C tile hold one
//     output channel each, so NHWC stores are 128-byte rows and GroupNorm per-channel partial sums
//     fall out of the epilogue with one cross-half exchange.
#include "hd_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvKArgs {
    const float* in0; const float* in1;
    int C0, C1, Cin;
    int B, H, W, IH, IW, stride, pad, upsample, KH, KW;
    const float* w; const float* bias;
    int Cout, CoutPad;
    int TB, TH, TW, LH, LW, npx, tiles_x, tiles_y, ntiles_n;
    int in_mode; const float* inA; const float* inB; const float* inE; int in_bstride;
    const float* ln_stats; const float* ln_g;
    int ep; const float* epScale; const float* epShift; int ep_bstride;
    float alpha; const float* res; const float* resA; const float* resB; int res_bstride;
    float* out; float* gn_part; int gn_slots;
};

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

template <int TM, int TN, int CK>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(ConvKArgs p) {
    constexpr int BM = 64 * TM, BN = 64 * TN, XS = CK + 1;
    constexpr int NW = (CK * BN / 4) / 256;           // float4 weight loads per thread per slab
    static_assert(NW >= 1, "slab too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int npx = p.npx;
    const int npx4 = (npx + 3) & ~3;
    int* pxsrc = reinterpret_cast<int*>(smem);        // [npx4] source pixel index or -1
    int* pxb = pxsrc + npx4;                          // [npx4] sample of that pixel
    int* rowpix = pxb + npx4;                         // [BM] output pixel index or -1
    int* rowb = rowpix + BM;                          // [BM]
    float* Ws = reinterpret_cast<float*>(rowb + BM);  // [2][CK][BN]
    float* Xs = Ws + 2 * CK * BN;                     // [npx][XS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;

    int bid = blockIdx.x;
    const int nt = bid % p.ntiles_n;
    int mt = bid / p.ntiles_n;
    const int tile_x = mt % p.tiles_x; mt /= p.tiles_x;
    const int tile_y = mt % p.tiles_y;
    const int tile_b = mt / p.tiles_y;
    const int b0 = tile_b * p.TB, y0 = tile_y * p.TH, x0 = tile_x * p.TW, n0 = nt * BN;
    const int LH = p.LH, LW = p.LW, thw = p.TH * p.TW, mvalid = p.TB * thw;

    for (int i = tid; i < npx; i += 256) {
        int tb = i / (LH * LW);
        int r = i - tb * LH * LW;
        int ly = r / LW, lx = r - ly * LW;
        int b = b0 + tb;
        int iy = y0 * p.stride + ly - p.pad, ix = x0 * p.stride + lx - p.pad;
        int src = -1;
        if (b < p.B) {
            if (p.upsample) {
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) src = (b * p.IH + (iy >> 1)) * p.IW + (ix >> 1);
            } else if (iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) {
                src = (b * p.IH + iy) * p.IW + ix;
            }
        }
        pxsrc[i] = src;
        pxb[i] = b < p.B ? b : p.B - 1;
    }
    for (int m = tid; m < BM; m += 256) {
        int tb = m / thw;
        int r = m - tb * thw;
        int ty = r / p.TW, tx = r - ty * p.TW;
        int b = b0 + tb, y = y0 + ty, x = x0 + tx;
        bool v = (m < mvalid) && b < p.B && y < p.H && x < p.W;
        rowpix[m] = v ? (b * p.H + y) * p.W + x : -1;
        rowb[m] = b < p.B ? b : p.B - 1;
    }

    int aoff[TM], boff[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        int m = wm * 32 * TM + tm * 32 + l31;
        if (m >= mvalid) m = 0;
        int tb = m / thw;
        int r = m - tb * thw;
        int ty = r / p.TW, tx = r - ty * p.TW;
        aoff[tm] = ((tb * LH + ty * p.stride) * LW + tx * p.stride) * XS + half;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) boff[tn] = half * BN + wn * 32 * TN + tn * 32 + l31;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int ntaps = p.KH * p.KW, nchunks = p.Cin / CK, nit = ntaps * nchunks;
    float4 wreg[NW];
    auto loadW = [&](int it) {
        int c = it / ntaps, tap = it - c * ntaps;
        const float* src = p.w + ((size_t)tap * p.Cin + (size_t)c * CK) * p.CoutPad + n0;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            int idx = tid + j * 256;
            int k = idx / (BN / 4), q = idx - k * (BN / 4);
            wreg[j] = *reinterpret_cast<const float4*>(src + (size_t)k * p.CoutPad + q * 4);
        }
    };
    auto storeW = [&](int buf) {
        float* dst = Ws + buf * CK * BN;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            int idx = tid + j * 256;
            int k = idx / (BN / 4), q = idx - k * (BN / 4);
            *reinterpret_cast<float4*>(dst + k * BN + q * 4) = wreg[j];
        }
    };

    loadW(0);
    for (int it = 0; it < nit; ++it) {
        const int c = it / ntaps, tap = it - c * ntaps;
        if (tap == 0) {
            __syncthreads();  // every wave is done reading the previous slice (and the tables are written)
            const int cc = c * CK;
            const float* src; int Csrc, coff;
            if (cc < p.C0) { src = p.in0; Csrc = p.C0; coff = cc; } else { src = p.in1; Csrc = p.C1; coff = cc - p.C0; }
            for (int i = tid; i < npx * (CK / 4); i += 256) {
                int px = i / (CK / 4), q = i - px * (CK / 4);
                int s = pxsrc[px];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (s >= 0) {
                    v = *reinterpret_cast<const float4*>(src + (size_t)s * Csrc + coff + q * 4);
                    if (p.in_mode == IN_AFFINE_SILU) {
                        const int o = pxb[px] * p.in_bstride + cc + q * 4;
                        const float4 A = *reinterpret_cast<const float4*>(p.inA + o);
                        const float4 Bv = *reinterpret_cast<const float4*>(p.inB + o);
                        v.x = silu_f(v.x * A.x + Bv.x); v.y = silu_f(v.y * A.y + Bv.y);
                        v.z = silu_f(v.z * A.z + Bv.z); v.w = silu_f(v.w * A.w + Bv.w);
                        if (p.inE) {
                            const float4 E = *reinterpret_cast<const float4*>(p.inE + o);
                            v.x += E.x; v.y += E.y; v.z += E.z; v.w += E.w;
                        }
                    } else if (p.in_mode == IN_LAYERNORM) {
                        const float mu = p.ln_stats[2 * (size_t)s], rs = p.ln_stats[2 * (size_t)s + 1];
                        const float4 g = *reinterpret_cast<const float4*>(p.ln_g + cc + q * 4);
                        v.x = (v.x - mu) * rs * g.x; v.y = (v.y - mu) * rs * g.y;
                        v.z = (v.z - mu) * rs * g.z; v.w = (v.w - mu) * rs * g.w;
                    }
                }
                float* d = Xs + px * XS + q * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
        storeW(it & 1);
        if (it + 1 < nit) loadW(it + 1);
        __syncthreads();
        const float* Wb = Ws + (it & 1) * CK * BN;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int tapoff = (ky * LW + kx) * XS;
#pragma unroll
        for (int kk = 0; kk < CK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm] = Xs[aoff[tm] + tapoff + 2 * kk];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = Wb[boff[tn] + 2 * kk * BN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * 32 * TN + tn * 32 + l31;
        const bool nok = n < p.Cout;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int pix = rowpix[m];
                if (pix < 0 || !nok) continue;
                float v = acc[tm][tn][r] + bias;
                s1 += v; s2 += v * v;
                if (p.ep & (EP_FILM_SILU | EP_ADD_SILU)) {
                    const int o = rowb[m] * p.ep_bstride + n;
                    if (p.ep & EP_FILM_SILU) v = v * (p.epScale[o] + 1.f) + p.epShift[o];
                    else v = v + p.epShift[o];
                    v = silu_f(v);
                }
                if (p.ep & EP_RES) v = p.alpha * v + p.res[(size_t)pix * p.Cout + n];
                if (p.ep & EP_RES_AFFINE_SILU) {
                    const int o = rowb[m] * p.res_bstride + n;
                    v += silu_f(p.res[(size_t)pix * p.Cout + n] * p.resA[o] + p.resB[o]);
                }
                p.out[(size_t)pix * p.Cout + n] = v;
            }
        }
        if (p.gn_part) {  // TB == 1: every row of this workgroup belongs to sample b0
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (half == 0 && nok) {
                const int slot = (tile_y * p.tiles_x + tile_x) * 2 + wm;
                float* d = p.gn_part + (((size_t)b0 * p.gn_slots + slot) * p.Cout + n) * 2;
                d[0] = s1; d[1] = s2;
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------

// Optional per-launch timing of the convolution kernels with HIP events on the launch stream
// (hd_profile_* in include/hicdiff_hip.h): bench.py's live roofline figure comes from here.
#include <vector>
struct ProfRec { int variant; double flops, bytes; hipEvent_t e0, e1; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
void hd_prof_enable(bool on) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
    g_prof_on = on;
}
// Sums per variant (0: BN=128 tile, 1: BN=64 tile). Synchronises on the recorded events.
void hd_prof_collect(double ms[2], double flops[2], double bytes[2], long long launches[2]) {
    for (int v = 0; v < 2; ++v) { ms[v] = flops[v] = bytes[v] = 0.0; launches[v] = 0; }
    for (auto& r : g_prof) {
        float t = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
            ms[r.variant] += t; flops[r.variant] += r.flops; bytes[r.variant] += r.bytes; launches[r.variant] += 1;
        }
    }
}

struct TileGeom { int TB, TH, TW; };

// Pick the TB x TH x TW output window (<= BM pixels) that wastes the fewest MFMA rows, then the
// fewest staged halo pixels.  Small feature maps take whole images (TB > 1).
#include <map>
#include <tuple>
static TileGeom pick_geom(int B, int H, int W, int BM, int KH, int KW, int stride, int max_px) {
    static std::map<std::tuple<int, int, int, int, int, int, int, int>, TileGeom> cache;
    const auto key = std::make_tuple(B, H, W, BM, KH, KW, stride, max_px);
    auto hit = cache.find(key);
    if (hit != cache.end()) return hit->second;
    TileGeom best{1, 1, 1};
    double best_score = -1.0;
    for (int tw = 1; tw <= W && tw <= BM; ++tw) {
        for (int th = 1; th <= H && th * tw <= BM; ++th) {
            int tb = 1;
            if (th == H && tw == W) { tb = BM / (H * W); if (tb > B) tb = B; if (tb < 1) tb = 1; }
            long lh = (long)(th - 1) * stride + KH, lw = (long)(tw - 1) * stride + KW;
            long npx = tb * lh * lw;
            if (npx > max_px) continue;
            long tiles = (long)((B + tb - 1) / tb) * ((H + th - 1) / th) * ((W + tw - 1) / tw);
            double eff = (double)B * H * W / ((double)tiles * BM);
            double halo = (double)npx / (double)(tb * th * tw * stride * stride);
            double score = eff - 0.02 * halo;
            if (score > best_score) { best_score = score; best = {tb, th, tw}; }
        }
    }
    cache[key] = best;
    return best;
}

int conv_gn_slots(int B, int H, int W, int Cout) {
    (void)Cout;
    TileGeom g = pick_geom(B, H, W, 128, 3, 3, 1, 512);
    if (g.TB != 1) return 0;
    return ((H + g.TH - 1) / g.TH) * ((W + g.TW - 1) / g.TW) * 2;
}

template <int TM, int TN, int CK>
static int launch_variant(ConvKArgs& k, hipStream_t st) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    const int npx4 = (k.npx + 3) & ~3;
    size_t lds = (size_t)(2 * npx4 + 2 * BM) * 4 + (size_t)2 * CK * BN * 4 + (size_t)k.npx * (CK + 1) * 4;
    if (lds > 64 * 1024) { hd_set_error("conv tile needs more than 64 KiB of LDS"); return -1; }
    const int mtiles = ((k.B + k.TB - 1) / k.TB) * k.tiles_y * k.tiles_x;
    k.ntiles_n = k.CoutPad / BN;
    dim3 grid((unsigned)(mtiles * k.ntiles_n));
    ProfRec rec{};
    if (g_prof_on) {
        rec.variant = TN == 2 ? 0 : 1;
        rec.flops = 2.0 * k.B * k.H * k.W * (double)k.Cout * k.Cin * k.KH * k.KW;
        rec.bytes = 4.0 * ((double)k.B * k.IH * k.IW * k.Cin + (double)k.B * k.H * k.W * k.Cout + (double)k.KH * k.KW * k.Cin * k.Cout);
        (void)hipEventCreate(&rec.e0); (void)hipEventCreate(&rec.e1);
        (void)hipEventRecord(rec.e0, st);
    }
    hipLaunchKernelGGL((conv_igemm_f32_kernel<TM, TN, CK>), grid, dim3(256), lds, st, k);
    if (g_prof_on) { (void)hipEventRecord(rec.e1, st); g_prof.push_back(rec); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("conv launch: ") + hipGetErrorString(e)); return -3; }
    return 0;
}

int launch_conv(const ConvArgs& a, hipStream_t st, int* gn_slots_out) {
    ConvKArgs k{};
    k.in0 = a.in0; k.in1 = a.in1; k.C0 = a.C0; k.C1 = a.C1; k.Cin = a.C0 + a.C1;
    k.B = a.B; k.H = a.H; k.W = a.W; k.IH = a.IH; k.IW = a.IW;
    k.stride = a.stride; k.pad = a.pad; k.upsample = a.upsample; k.KH = a.cw.KH; k.KW = a.cw.KW;
    k.w = a.cw.w; k.bias = a.cw.bias; k.Cout = a.cw.Cout; k.CoutPad = a.cw.CoutPad;
    k.in_mode = a.in_mode; k.inA = a.inA; k.inB = a.inB; k.inE = a.inE; k.in_bstride = a.in_bstride;
    k.ln_stats = a.ln_stats; k.ln_g = a.ln_g;
    k.ep = a.ep; k.epScale = a.epScale; k.epShift = a.epShift; k.ep_bstride = a.ep_bstride;
    k.alpha = a.alpha; k.res = a.res; k.resA = a.resA; k.resB = a.resB; k.res_bstride = a.res_bstride;
    k.out = a.out;
    if (k.Cin != a.cw.Cin || k.Cin % 16 != 0 || (a.C1 && a.C0 % 16 != 0) || k.CoutPad % 64 != 0) {
        hd_set_error("conv: channel counts must be multiples of 16 and match the packed weight");
        return -1;
    }
    const int BM = 128;
    TileGeom g = pick_geom(a.B, a.H, a.W, BM, k.KH, k.KW, a.stride, 512);
    k.TB = g.TB; k.TH = g.TH; k.TW = g.TW;
    k.LH = (g.TH - 1) * a.stride + k.KH; k.LW = (g.TW - 1) * a.stride + k.KW;
    k.npx = g.TB * k.LH * k.LW;
    k.tiles_y = (a.H + g.TH - 1) / g.TH; k.tiles_x = (a.W + g.TW - 1) / g.TW;
    k.gn_part = nullptr; k.gn_slots = 0;
    if (a.gn_part && g.TB == 1) { k.gn_part = a.gn_part; k.gn_slots = k.tiles_y * k.tiles_x * 2; }
    if (gn_slots_out) *gn_slots_out = k.gn_slots;
    if (k.CoutPad % 128 == 0) return launch_variant<2, 2, 16>(k, st);
    return launch_variant<2, 1, 16>(k, st);
}

// ---- weight packing ------------------------------------------------------------------------------
// src: torch layout [Cout][Cin][KH][KW]; dst: [KH*KW][Cin][CoutPad] (pad columns zeroed by the
// caller's memset).  standardize: (w - mean_o) * rsqrt(var_o + 1e-5), biased variance over
// (Cin,KH,KW), src/hicdiff.py:89-97.  unshuffle: src is the 1x1 weight [Cout][4*C] applied after
// 'b c (h p1) (w p2) -> b (c p1 p2) h w' (src/hicdiff.py:80); it becomes a 2x2 stride-2 conv with
// tap = p1*2 + p2 and cin = c.
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout,
                                                        int Cin, int KH, int KW, int CoutPad, int standardize,
                                                        int unshuffle) {
    const int o = blockIdx.x;
    const int n = Cin * KH * KW;   // elements of this filter (for unshuffle: Cin*4 with KH=KW=2)
    const float* s = src + (size_t)o * n;
    __shared__ double red[256];
    double mean = 0.0, rstd = 1.0;
    if (standardize) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) acc += s[i];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
        mean = red[0] / n;
        __syncthreads();
        acc = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) { double d = s[i] - mean; acc += d * d; }
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
        rstd = 1.0 / sqrt(red[0] / n + 1e-5);
    }
    const int taps = KH * KW;
    for (int i = threadIdx.x; i < n; i += 256) {
        int cin, tap;
        if (unshuffle) { cin = i / 4; tap = i % 4; }          // i = c*4 + p1*2 + p2
        else { cin = i / taps; tap = i % taps; }              // i = c*KH*KW + ky*KW + kx
        dst[((size_t)tap * Cin + cin) * CoutPad + o] = (float)(((double)s[i] - mean) * rstd);
    }
}

int launch_pack_conv(const float* src, float* dst, int Cout, int Cin, int KH, int KW, int CoutPad, int standardize,
                     int unshuffle, hipStream_t st) {
    hipError_t e = hipMemsetAsync(dst, 0, (size_t)KH * KW * Cin * CoutPad * sizeof(float), st);
    if (e != hipSuccess) { hd_set_error("pack memset failed"); return -3; }
    hipLaunchKernelGGL(pack_conv_kernel, dim3(Cout), dim3(256), 0, st, src, dst, Cout, Cin, KH, KW, CoutPad, standardize,
                       unshuffle);
    return 0;
}

// dst[c][dst_col0 + r] = src[r][c]  (torch Linear weight [out][in] -> [in][out_total] slab)
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols, int dst_ld,
                                 int dst_col0) {
    __shared__ float t[32][33];
    int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        int r = r0 + j, c = c0 + threadIdx.x;
        t[j][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        int c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) dst[(size_t)c * dst_ld + dst_col0 + r] = t[threadIdx.x][j];
    }
}

int launch_transpose(const float* src, float* dst, int rows, int cols, int dst_ld, int dst_col0, hipStream_t st) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, st, src, dst, rows, cols, dst_ld, dst_col0);
    return 0;
}
