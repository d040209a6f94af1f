// Split-bf16 x3 implicit-GEMM convolution on v_mfma_f32_32x32x16_bf16 -- the default arithmetic.
//
// Every fp32 operand is split into two bf16 values, x = hi + lo (16 significant bits), and a product
// is evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation: three MFMAs at 16x the fp32-MFMA rate.
// Dropped terms are O(2^-17) relative: ~2e-5 error on an epsilon forward, 1e-4 on a 50-step chain
// (plain bf16 operands give 1e-2 and fail the 1e-3 parity bound).
//
// LDS rows (one staged pixel of X / one output channel of W) are [CK bf16 hi | CK bf16 lo | 16 B pad];
// the pitch 4*CK+16 bytes is an odd multiple of 16, so the 16 lanes of a ds_read_b128 group (16
// consecutive pixels / channels) hit 16 different 16-byte bank slots.  Weights arrive pre-split
// ([tap][Cin/CK][CoutPad][hi|lo], packed at load time) and are copied 16 bytes per lane; activations
// are split on the fly while the loader applies its transform.
//
// Pipeline per workgroup (4 waves): the raw fp32 values of the NEXT K-slice's input window are
// prefetched into registers right after the current slice is staged, so their HBM/L2 latency is
// covered by the slice's KH*KW tap iterations; the weight slab of the next tap is prefetched into
// registers during the current tap's MFMAs and written to the other LDS buffer afterwards: one
// barrier per tap, two at a slice switch.
#include "conv_device.h"

#include <set>

// Diagnostic build (-DHD_STAMP): s_memtime stamps around the phases of the main loop; never quote its run
// time, only the shares (cdna_hip_programming.md section 7, In-kernel stamps).
#ifdef HD_STAMP
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned short h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 hh = (__bf16)v[j];
        const __bf16 ll = (__bf16)(v[j] - (float)hh);
        h[j] = __builtin_bit_cast(unsigned short, hh);
        l[j] = __builtin_bit_cast(unsigned short, ll);
    }
    hi = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
    lo = make_uint4(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16), l[4] | ((unsigned)l[5] << 16), l[6] | ((unsigned)l[7] << 16));
}

// loader transform specialised at compile time (MODE = IN_NONE / IN_AFFINE_SILU / IN_LAYERNORM) so the
// staging code is branch-free
template <int MODE>
__device__ __forceinline__ float4 xform4(const ConvKArgs& p, float4 v, int cc, int b, float mu, float rs) {
    if constexpr (MODE == IN_AFFINE_SILU) {
        const int o = b + cc;                         // b = sample * in_bstride
        const float4 A = *reinterpret_cast<const float4*>(p.inA + o);
        const float4 Bv = *reinterpret_cast<const float4*>(p.inB + o);
        v.x = silu_f(v.x * A.x + Bv.x); v.y = silu_f(v.y * A.y + Bv.y);
        v.z = silu_f(v.z * A.z + Bv.z); v.w = silu_f(v.w * A.w + Bv.w);
        if (p.inE) {   // wave-uniform (kernel argument)
            const float4 E = *reinterpret_cast<const float4*>(p.inE + o);
            v.x += E.x; v.y += E.y; v.z += E.z; v.w += E.w;
        }
    } else if constexpr (MODE == IN_LAYERNORM) {
        // s carries nothing here: the per-pixel statistics are loaded once per item (they do not depend on
        // the K slice) and passed in through (mu, rs)
        const float4 g = *reinterpret_cast<const float4*>(p.ln_g + cc);
        v.x = (v.x - mu) * rs * g.x; v.y = (v.y - mu) * rs * g.y;
        v.z = (v.z - mu) * rs * g.z; v.w = (v.w - mu) * rs * g.w;
    }
    return v;
}

// WM x WN waves, each TM x TN accumulator tiles of 32 x 32; MAXI = staged 8-channel items per thread.
template <int WM, int WN, int TM, int TN, int CK, int MAXI, int MODE>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_bf16x3_kernel(ConvKArgs p) {
    constexpr int NT = 64 * WM * WN;                   // 4 waves (256 threads) or 8 waves (512 threads)
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, PITCH = 4 * CK + 16, ROWB = 4 * CK;
    constexpr int KS = CK / 16;                        // k16 MFMA steps per slab
    constexpr int IPP = CK / 8;                        // 8-channel items per staged pixel
    constexpr int NW = (BN * (CK / 4)) / NT;           // 16-byte weight pieces per thread per slab
    static_assert(NW >= 1 && (NT == 256 || NT == 512), "bad tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int npx = p.npx;
    const int npx4 = (npx + 3) & ~3;
    int* pxsrc = reinterpret_cast<int*>(smem);
    int* pxb = pxsrc + npx4;
    int* rowpix = pxb + npx4;
    int* rowb = rowpix + BM;
    char* Ws = reinterpret_cast<char*>(rowb + BM);     // [2][BN][PITCH]
    char* Xs = Ws + 2 * BN * PITCH;                    // [npx][PITCH]

    const TileCtx t = tile_decode<WN, BN>(p);
    const int tid = t.tid;
    init_tables<BM, NT>(p, t, pxsrc, pxb, rowpix, rowb);

    int aoff[TM], boff[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) aoff[tm] = row_px_offset<TM>(p, t, tm) * PITCH + t.half * 16;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) boff[tn] = (t.wn * 32 * TN + tn * 32 + t.l31) * PITCH + t.half * 16;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int ntaps = p.KH * p.KW, nchunks = p.Cin / CK, nit = ntaps * nchunks;
    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0, acc_w = 0, acc_b = 0, acc_m = 0, acc_s = 0, t_begin = 0;
    (void)tk0; (void)tk1; (void)tk2; (void)tk3; (void)tk4; (void)acc_w; (void)acc_b; (void)acc_m; (void)acc_s; (void)t_begin;
    STAMP(t_begin);
    __syncthreads();                 // tables visible

    // per-thread staging items: item i = tid + 256*j = (pixel i / IPP, channels 8*(i % IPP)..+7).  256 is a
    // multiple of IPP, so the channel offset q8 is the same for all of a thread's items and its pixels are
    // PXSTEP apart.  Source indices are clamped so every lane issues every load (no divergent control flow
    // around memory operations); padding pixels (src < 0) and items past the window are neutralised when
    // the LDS rows are written.
    constexpr int PXSTEP = NT / IPP;
    const int q8 = (tid % IPP) * 8, px0 = tid / IPP;
    int it_src[MAXI], it_aff[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int px = px0 + j * PXSTEP;
        const int pxc = px < npx ? px : npx - 1;
        it_src[j] = pxsrc[pxc];                                  // < 0: zero padding
        it_aff[j] = MODE == IN_AFFINE_SILU ? pxb[pxc] * p.in_bstride : 0;
    }
    float ln_mu[MAXI], ln_rs[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        ln_mu[j] = 0.f; ln_rs[j] = 1.f;
        if constexpr (MODE == IN_LAYERNORM) {
            const float2 st = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (size_t)(it_src[j] < 0 ? 0 : it_src[j]));
            ln_mu[j] = st.x; ln_rs[j] = st.y;
        }
    }

    // weight-slab prefetch registers: named scalars (an indexed array here is left in scratch by hipcc)
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0;
    static_assert(NW <= 4, "weight slab prefetch holds at most four 16-byte pieces per thread");
#define HD_WLOAD(k)                                                                                        \
    if constexpr (NW > k) {                                                                                \
        const int idx = tid + k * NT;                                                                      \
        const int row = idx / (CK / 4), piece = idx - row * (CK / 4);                                      \
        w##k = *reinterpret_cast<const uint4*>(wsrc + (size_t)row * ROWB + piece * 16);                    \
    }
#define HD_WSTORE(k)                                                                                       \
    if constexpr (NW > k) {                                                                                \
        const int idx = tid + k * NT;                                                                      \
        const int row = idx / (CK / 4), piece = idx - row * (CK / 4);                                      \
        *reinterpret_cast<uint4*>(dst + row * PITCH + piece * 16) = w##k;                                  \
    }
    float4 xr[MAXI][2];

    // ---- prologue: raw values of slice 0, weight slab of (slice 0, tap 0)
    {
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            const float* g = p.in0 + (size_t)(it_src[j] < 0 ? 0 : it_src[j]) * p.C0 + q8;
            xr[j][0] = xr[j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(p.ablate & 32)) {
                xr[j][0] = *reinterpret_cast<const float4*>(g);
                xr[j][1] = *reinterpret_cast<const float4*>(g + 4);
            }
        }
        const char* wsrc = reinterpret_cast<const char*>(p.wsplit) + (size_t)t.n0 * ROWB;
        HD_WLOAD(0) HD_WLOAD(1) HD_WLOAD(2) HD_WLOAD(3)
    }

    int buf = 0;
    for (int c = 0; c < nchunks; ++c) {
        STAMP(tk4);
        if (c > 0) __syncthreads();      // every wave is done reading the previous slice's window
        // ---- stage slice c: transform + split + LDS write of the prefetched raw values
        if (!(p.ablate & 64)) {
            const int cc = c * CK + q8;
#pragma unroll
            for (int j = 0; j < MAXI; ++j) {
                const float4 v0 = xform4<MODE>(p, xr[j][0], cc, it_aff[j], ln_mu[j], ln_rs[j]);
                const float4 v1 = xform4<MODE>(p, xr[j][1], cc + 4, it_aff[j], ln_mu[j], ln_rs[j]);
                uint4 hi, lo;
                split8(v0, v1, hi, lo);
                if (it_src[j] < 0) { hi = make_uint4(0, 0, 0, 0); lo = hi; }   // zero padding is applied AFTER the transform
                const int px = px0 + j * PXSTEP;
                if (px < npx && !(p.ablate & 2)) {
                    char* d = Xs + px * PITCH + q8 * 2;
                    *reinterpret_cast<uint4*>(d) = hi;
                    *reinterpret_cast<uint4*>(d + 2 * CK) = lo;
                }
            }
        }
        // The next slice's raw values are requested one item per tap, AFTER that tap's weight loads (below):
        // vmcnt retires in issue order, so a wait for the weight slab never has to drain a younger, slower
        // activation load, and every activation load gets about two tap iterations to land.
        const float* nsrc = nullptr; int nCsrc = 0, ncoff = 0;
        if (c + 1 < nchunks) {
            const int cc = (c + 1) * CK;
            if (cc < p.C0) { nsrc = p.in0; nCsrc = p.C0; ncoff = cc; } else { nsrc = p.in1; nCsrc = p.C1; ncoff = cc - p.C0; }
        }
        STAMP(tk0);
#ifdef HD_STAMP
        acc_s += tk0 - tk4;
#endif
        int ky = 0, kx = 0;
        for (int tap = 0; tap < ntaps; ++tap) {
            STAMP(tk0);
            // weight slab (c, tap) -> LDS; prefetch the next slab into registers
            {
                char* dst = Ws + buf * BN * PITCH;
                if (!(p.ablate & 4)) { HD_WSTORE(0) HD_WSTORE(1) HD_WSTORE(2) HD_WSTORE(3) }
                int tapn = tap + 1, cn = c;
                if (tapn == ntaps) { tapn = 0; cn = c + 1; }
                if (cn < nchunks && !(p.ablate & 256)) {
                    const char* wsrc = reinterpret_cast<const char*>(p.wsplit) + ((size_t)(tapn * nchunks + cn) * p.CoutPad + t.n0) * ROWB;
                    HD_WLOAD(0) HD_WLOAD(1) HD_WLOAD(2) HD_WLOAD(3)
                }
                if (nsrc && !(p.ablate & 32)) {
                    // items tap*IPT .. of the next slice (all remaining ones on the last tap)
                    constexpr int IPT = (MAXI + 8) / 9;          // items per tap when there are 9 taps
#pragma unroll
                    for (int j = 0; j < MAXI; ++j) {
                        const bool mine = (ntaps >= 9) ? (j / IPT == tap || (tap == ntaps - 1 && j / IPT >= ntaps)) : (tap == ntaps - 1);
                        if (mine) {
                            const float* g = nsrc + (size_t)(it_src[j] < 0 ? 0 : it_src[j]) * nCsrc + ncoff + q8;
                            xr[j][0] = *reinterpret_cast<const float4*>(g);
                            xr[j][1] = *reinterpret_cast<const float4*>(g + 4);
                        }
                    }
                }
            }
            STAMP(tk1);
            __syncthreads();             // X window (if just staged) and weight slab visible
            STAMP(tk2);
            const char* Wb = Ws + buf * BN * PITCH;
            const int tapoff = (ky * p.LW + kx) * PITCH;
            if (!(p.ablate & 8))
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const char* a = Xs + aoff[tm] + tapoff + s * 32;
                    ah[tm] = *reinterpret_cast<const bf16x8*>(a);
                    al[tm] = *reinterpret_cast<const bf16x8*>(a + 2 * CK);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const char* b = Wb + boff[tn] + s * 32;
                    bh[tn] = *reinterpret_cast<const bf16x8*>(b);
                    bl[tn] = *reinterpret_cast<const bf16x8*>(b + 2 * CK);
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    }
            }
            STAMP(tk3);
#ifdef HD_STAMP
            acc_w += tk1 - tk0; acc_b += tk2 - tk1; acc_m += tk3 - tk2;
#endif
            buf ^= 1;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    }
    (void)nit; (void)w1; (void)w2; (void)w3;
#undef HD_WLOAD
#undef HD_WSTORE
    STAMP(tk0);
    conv_epilogue<BM, BN, TM, TN, NT>(p, t, acc, rowpix, rowb, reinterpret_cast<float*>(Ws));
#ifdef HD_STAMP
    STAMP(tk1);
    if (p.stamp && t.lane == 0) {
        atomicAdd(p.stamp + 0, acc_w); atomicAdd(p.stamp + 1, acc_b); atomicAdd(p.stamp + 2, acc_m); atomicAdd(p.stamp + 3, acc_s);
        atomicAdd(p.stamp + 4, tk1 - tk0); atomicAdd(p.stamp + 5, tk1 - t_begin); atomicAdd(p.stamp + 6, 1ull); atomicAdd(p.stamp + 7, (unsigned long long)nit);
    }
#endif
}

template <typename K>
static int launch_one(K kernel, ConvLaunch& L, hipStream_t st, int nthreads = 256) {
    static std::set<const void*> raised;   // every instantiation has the same pointer TYPE: key by address
    if (!raised.count(reinterpret_cast<const void*>(kernel))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hd_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); return -3;
        }
        raised.insert(reinterpret_cast<const void*>(kernel));
    }
    const ConvKArgs& k = L.k;
    const int mtiles = ((k.B + k.TB - 1) / k.TB) * k.tiles_y * k.tiles_x;
    dim3 grid((unsigned)(mtiles * k.ntiles_n));
    conv_prof_begin(L, st);
    hipLaunchKernelGGL(kernel, grid, dim3(nthreads), L.lds, st, L.k);
    conv_prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("conv launch: ") + hipGetErrorString(e)); return -3; }
    return 0;
}

// MAXI bounds npx: 256 * MAXI / (CK / 8) staged pixels (conv_host.hip plans inside it).
template <int MODE>
static int launch_mode(ConvLaunch& L, hipStream_t st) {
    if (L.ck == 32) {
        if (L.cfg == 0) return launch_one(conv_igemm_bf16x3_kernel<2, 2, 2, 2, 32, 5, MODE>, L, st);
        if (L.cfg == 1) return launch_one(conv_igemm_bf16x3_kernel<2, 2, 2, 1, 32, 5, MODE>, L, st);
        if (L.cfg == 3) return launch_one(conv_igemm_bf16x3_kernel<4, 2, 2, 2, 32, 3, MODE>, L, st, 512);
        return launch_one(conv_igemm_bf16x3_kernel<4, 1, 2, 2, 32, 6, MODE>, L, st);
    }
    if (L.cfg == 0) return launch_one(conv_igemm_bf16x3_kernel<2, 2, 2, 2, 16, 3, MODE>, L, st);
    if (L.cfg == 1) return launch_one(conv_igemm_bf16x3_kernel<2, 2, 2, 1, 16, 3, MODE>, L, st);
    if (L.cfg == 3) return launch_one(conv_igemm_bf16x3_kernel<4, 2, 2, 2, 16, 2, MODE>, L, st, 512);
    return launch_one(conv_igemm_bf16x3_kernel<4, 1, 2, 2, 16, 3, MODE>, L, st);
}

int launch_conv_bf16x3(ConvLaunch& L, hipStream_t st) {
    if (L.k.in_mode == IN_AFFINE_SILU) return launch_mode<IN_AFFINE_SILU>(L, st);
    if (L.k.in_mode == IN_LAYERNORM) return launch_mode<IN_LAYERNORM>(L, st);
    return launch_mode<IN_NONE>(L, st);
}
