// Exact-fp32 implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain; 157 TFLOP/s
// peak): the strict-parity arithmetic of the engine (hd_set_precision(ctx, HD_PRECISION_F32)).
// Structure: see conv_host.hip.  LDS: X window [npx][CK+1] fp32 (odd pitch: conflict-free column
// reads), weight slab [2][CK][BN] fp32.
#include "conv_device.h"

#include <set>

template <int TM, int TN, int CK>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(ConvKArgs p) {
    constexpr int WM = 2, WN = 2;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, XS = CK + 1;
    constexpr int NW = (CK * BN / 4) / 256;           // float4 weight loads per thread per slab
    static_assert(NW >= 1, "slab too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int npx = p.npx;
    const int npx4 = (npx + 3) & ~3;
    int* pxsrc = reinterpret_cast<int*>(smem);        // [npx4]
    int* pxb = pxsrc + npx4;                          // [npx4]
    int* rowpix = pxb + npx4;                         // [BM]
    int* rowb = rowpix + BM;                          // [BM]
    float* Ws = reinterpret_cast<float*>(rowb + BM);  // [2][CK][BN]
    float* Xs = Ws + 2 * CK * BN;                     // [npx][XS]

    const TileCtx t = tile_decode<WN, BN>(p);
    const int tid = t.tid;
    init_tables<BM, 256>(p, t, pxsrc, pxb, rowpix, rowb);

    int aoff[TM], boff[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) aoff[tm] = row_px_offset<TM>(p, t, tm) * XS + t.half;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) boff[tn] = t.half * BN + t.wn * 32 * TN + tn * 32 + t.l31;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int ntaps = p.KH * p.KW, nchunks = p.Cin / CK, nit = ntaps * nchunks;
    float4 wreg[NW];
    auto loadW = [&](int c, int tap) {
        const float* src = p.w + ((size_t)tap * p.Cin + (size_t)c * CK) * p.CoutPad + t.n0;
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

    loadW(0, 0);
    int c = 0, tap = 0, ky = 0, kx = 0;
    for (int it = 0; it < nit; ++it) {
        if (tap == 0) {
            __syncthreads();  // every wave is done reading the previous slice (and the tables are written)
            const int cc = c * CK;
            const float* src; int Csrc, coff;
            if (cc < p.C0) { src = p.in0; Csrc = p.C0; coff = cc; } else { src = p.in1; Csrc = p.C1; coff = cc - p.C0; }
            for (int i = tid; i < npx * (CK / 4); i += 256) {
                int px = i / (CK / 4), q = i - px * (CK / 4);
                int s = pxsrc[px];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (s >= 0) v = transform4(p, *reinterpret_cast<const float4*>(src + (size_t)s * Csrc + coff + q * 4), cc + q * 4, s, pxb[px]);
                float* d = Xs + px * XS + q * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
        storeW(it & 1);
        int tapn = tap + 1, cn = c;
        if (tapn == ntaps) { tapn = 0; cn = c + 1; }
        if (it + 1 < nit) loadW(cn, tapn);
        __syncthreads();
        const float* Wb = Ws + (it & 1) * CK * BN;
        const int tapoff = (ky * p.LW + kx) * XS;
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
        tap = tapn; c = cn;
        if (++kx == p.KW) { kx = 0; ++ky; }
        if (tap == 0) { ky = 0; kx = 0; }
    }
    conv_epilogue<BM, BN, TM, TN, 256>(p, t, acc, rowpix, rowb, Ws);
}

template <typename K>
static int launch_one(K kernel, const char* name, ConvLaunch& L, hipStream_t st) {
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
    conv_prof_begin(L, name, st);
    hipLaunchKernelGGL(kernel, grid, dim3(256), L.lds, st, L.k);
    conv_prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("conv launch: ") + hipGetErrorString(e)); return -3; }
    return 0;
}

int launch_conv_f32(ConvLaunch& L, hipStream_t st) {
    if (L.cfg == 0) return launch_one(conv_igemm_f32_kernel<2, 2, 16>, conv_prof_name("conv_igemm_f32_kernel<2, 2, 16>", -1, ""), L, st);
    return launch_one(conv_igemm_f32_kernel<2, 1, 16>, conv_prof_name("conv_igemm_f32_kernel<2, 1, 16>", -1, ""), L, st);
}
