// Split-bf16 x3 implicit-GEMM convolution on v_mfma_f32_32x32x16_bf16 -- the default arithmetic.
//
// Every fp32 operand is split into two bf16 values, x = hi + lo (16 significant bits), and a product
// is evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation: three MFMAs at 16x the fp32-MFMA rate.
// Dropped terms are O(2^-17) relative: ~2e-5 error on an epsilon forward, 1e-4 on a 50-step chain
// (plain bf16 operands give 1e-2 and fail the 1e-3 parity bound).
//
// LDS images.  Activations: one row per staged pixel, [CK bf16 hi | CK bf16 lo | 16 B pad]; the pitch 4*CK+16 bytes is an odd multiple
// of 16, so the 16 lanes of a ds_read_b128 group (16 consecutive pixels) hit 16 different 16-byte bank slots.  They are split on the fly
// while the loader applies its transform.  Weights: the global image is ALWAYS packed in 16-channel steps,
// [tap][Cin/16][CoutPad][16 hi | 16 lo] (64-byte rows, split once at load time), whatever the kernel's activation slice CK is, so the
// weights of one MFMA k-step -- a "k16-slab", BN rows of 64 bytes -- are one contiguous block.  In LDS a k16-slab is UNPADDED: piece pc
// (16 bytes) of row n sits at n*64 + ((pc ^ ((n >> 2) & 3)) << 4); the rows a ds_read_b128 lane group touches are distinct modulo 16, so
// the XOR spreads them over the 16 bank slots (checked for the 32x32x16 and the 16x16x32 operand maps).  The workgroup keeps a RING of
// 2*KS k16-slabs (KS = CK/16 k-steps per tap): the same bytes as two padded slabs took, minus the padding.
//
// Pipeline of the 3x3 kernels.  One iteration per (K slice c, filter tap), ONE barrier each, fragments read ONE K-STEP AHEAD:
//   at the barrier of iteration `it` every wave already holds the operand fragments of the tap's first k-step in registers (they were
//   read from LDS under the MFMAs of iteration it-1; only the window fragments of a slice's first tap cannot be, the window is staged at
//   the slice switch).  After the barrier, in program order:
//   1. the weight unit U(it) = k16-slabs KS*it + KS+1 .. KS*it + 2*KS goes from its prefetch registers to the ring slots whose previous
//      tenants (k16-slabs <= KS*it) every wave has finished reading -- the barrier's lgkmcnt(0) saw to that -- and unit U(it + WD) is
//      requested into the registers just freed (three register sets in flight: WD = 3);
//   2. this tap's share of the raw fp32 values of slice c+1 is requested (registers xr);
//   3. per k-step: the NEXT step's fragments are requested (ds_read_b128 into the second fragment set; for a tap's last step they belong
//      to the next tap: its k16-slab was written during iteration it-1 and became visible at this barrier), then the 12 (6 with TN = 1)
//      MFMAs of the current step run on fragments that have been in registers since the previous step.
// So no MFMA waits for an LDS round trip issued in its own step, and everything in 1 and 2 is asynchronous.  The loop body is
// STRAIGHT-LINE code (taps unrolled, last slice peeled, no data-dependent branch): hipcc places exact s_waitcnt vmcnt(N) only in
// straight-line code -- with branches between a load and its use it falls back to vmcnt(0), which serialises every load with the MFMA
// cluster (measured: the kernel then runs at the SUM of its load, staging and MFMA times).
// The any-filter kernels (1x1, 2x2 stride 2, LayerNorm / softmax loaders: taps in a runtime loop) keep the simpler order: slab it+1 is
// written during iteration it, fragments are read in the step that uses them.
// The per-slice operands of the loader transform (GroupNorm/FiLM scale and shift, LayerNorm gain) come from
// a small double-buffered LDS table filled one slice ahead, so staging a slice touches no global memory.
// Loader transforms: none, GroupNorm-apply + FiLM + SiLU (+ additive term), channel LayerNorm, softmax over each
// 32-channel head (LinearAttention's q); weights may be per sample (one sample per tile).
// 8-wave variant (one workgroup per CU): the activation window is double-buffered and slice c+1 is
// transformed/split/written item by item during the taps of slice c; 4-wave variants (two workgroups per
// CU, which overlap each other) keep one window and stage between two barriers at the slice switch.
#pragma once
#include "conv_device.h"

#include <set>
#include <type_traits>

// Diagnostic switches (HICDIFF_ABLATE, make DIAG=1): timing experiments only.  Compiled out of the product
// build, where their branches would break the straight-line loop body.
#ifdef HD_DIAG
#define ABL(bit) (p.ablate & (bit))
#else
#define ABL(bit) false
#endif

// make EXTRA=-DHD_STAMPS: wave 0 of every workgroup below 4096 leaves cycle stamps (start, loop entry, loop exit, end, cycles spent in the
// loop's barriers, XCC / CU id) in a device array that tools/conv_probe.py prints -- a timing study, never part of the product build.
#ifdef HD_STAMPS
// p.stamps: [4096][16] device words shared by every translation unit (conv_host.hip owns it).  [6]: cycles from kernel entry to the first
// window's loads being issued; [7]: from there until they have arrived (s_waitcnt vmcnt(0)); [12..14]: the epilogue's three figures
#define HD_STAMP() __builtin_readcyclecounter()
#endif

enum { IN_AFFINE_SILU_E = 3 };   // kernel-side mode: IN_AFFINE_SILU with the additive term inE (SR3 blocks)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

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

// eight fp32 -> eight fp16 (round to nearest even), packed like split8's halves
__device__ __forceinline__ uint4 half8(const float4& a, const float4& b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    auto pk = [](float x, float y) { const h2 v = {(_Float16)x, (_Float16)y}; return __builtin_bit_cast(unsigned, v); };
    return make_uint4(pk(a.x, a.y), pk(a.z, a.w), pk(b.x, b.y), pk(b.z, b.w));
}

// loader transform on 4 channels; pt: this item's entry of the slice's parameter table in LDS
// ([vector][sample][CK] floats; vectors: scale, shift(, additive) or the LayerNorm gain), ptv: bytes per vector
template <int MODE>
__device__ __forceinline__ float4 xform4(float4 v, const char* pt, int ptv, float mu, float rs) {
    if constexpr (MODE == IN_AFFINE_SILU || MODE == IN_AFFINE_SILU_E) {
        const float4 A = *reinterpret_cast<const float4*>(pt);
        const float4 Bv = *reinterpret_cast<const float4*>(pt + ptv);
        v.x = silu_f(v.x * A.x + Bv.x); v.y = silu_f(v.y * A.y + Bv.y);
        v.z = silu_f(v.z * A.z + Bv.z); v.w = silu_f(v.w * A.w + Bv.w);
        if constexpr (MODE == IN_AFFINE_SILU_E) {
            const float4 E = *reinterpret_cast<const float4*>(pt + 2 * ptv);
            v.x += E.x; v.y += E.y; v.z += E.z; v.w += E.w;
        }
    } else if constexpr (MODE == IN_LAYERNORM) {
        const float4 g = *reinterpret_cast<const float4*>(pt);
        v.x = (v.x - mu) * rs * g.x; v.y = (v.y - mu) * rs * g.y;
        v.z = (v.z - mu) * rs * g.z; v.w = (v.w - mu) * rs * g.w;
    }
    return v;
}

// x over the four lanes of a quad (the four 8-channel items of one pixel's 32-channel slice): DPP quad_perm, no LDS
__device__ __forceinline__ float quad_xor1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false)); }
__device__ __forceinline__ float quad_xor2(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false)); }

// IN_SOFTMAX32: softmax over the 32 channels of the slice (LinearAttention's q.softmax(dim=-2), src/hicdiff.py:217;
// one head = one K slice).  v0|v1 are this lane's 8 channels, the other 24 sit in the three neighbouring lanes.
__device__ __forceinline__ void softmax32(float4& v0, float4& v1) {
    float m = fmaxf(fmaxf(fmaxf(v0.x, v0.y), fmaxf(v0.z, v0.w)), fmaxf(fmaxf(v1.x, v1.y), fmaxf(v1.z, v1.w)));
    m = fmaxf(m, quad_xor1(m)); m = fmaxf(m, quad_xor2(m));
    v0.x = __expf(v0.x - m); v0.y = __expf(v0.y - m); v0.z = __expf(v0.z - m); v0.w = __expf(v0.w - m);
    v1.x = __expf(v1.x - m); v1.y = __expf(v1.y - m); v1.z = __expf(v1.z - m); v1.w = __expf(v1.w - m);
    float s = (v0.x + v0.y) + (v0.z + v0.w) + (v1.x + v1.y) + (v1.z + v1.w);
    s += quad_xor1(s); s += quad_xor2(s);
    const float r = 1.f / s;
    v0.x *= r; v0.y *= r; v0.z *= r; v0.w *= r; v1.x *= r; v1.y *= r; v1.z *= r; v1.w *= r;
}

// WM x WN waves, each TM x TN accumulator tiles of 32 x 32; MAXI = staged 8-channel items per thread;
// NTAPS = 9: 3x3 filter, taps unrolled; NTAPS = 0: any filter (1x1, 2x2 stride 2, ...), taps in a loop.
// AR (arithmetic of a product): 0 = split-bf16 x3 (hi hi + hi lo + lo hi); 1 = PLAIN, one bf16 MFMA (hi x hi only) -- the optional bf16
// arithmetic of the training step; 2 = F16W2 (round 4), two fp16 MFMAs xh (wh + wl): the activation is rounded ONCE to fp16 (11 significant
// bits), the weight image holds fp16 hi | lo of the exact weight -- a third fewer MFMAs and no lo half of the window, for the timestep band
// whose error the chain damps (tests/studies/error_budget_study.py); 3 = F16W1, ONE fp16 MFMA xh wh (both operands
// rounded once; the hi half of the same fp16 weight image) for the earliest quarter of a long chain.  The staging layout is the same in all.
#ifndef HD_CK16_CAP3
#define HD_CK16_CAP3 1    // 1: the plain-loader 256 x 64 tile with 16-channel slices is capped at 168 registers (three workgroups per CU)
#endif
template <int WM, int WN, int CK, int MODE, int NTAPS>
constexpr bool conv_cap3() { return HD_CK16_CAP3 && NTAPS == 9 && CK == 16 && WM == 4 && WN == 1 && MODE == IN_NONE; }
#ifndef HD_CONV_XD2
#define HD_CONV_XD2 1     // any-filter kernels: two activation prefetch register sets (0: one, the round-2 form; A/B builds)
#endif
template <int WM, int WN, int TM, int TN, int CK, int MAXI, int MODE, int NTAPS, int AR, bool FBWD = false>
__device__ __forceinline__ void conv_igemm_bf16x3_body(ConvKArgs p) {
    constexpr bool F16 = AR >= 2;
    constexpr bool ALO = AR == 0;                      // the window carries a lo half
    constexpr bool BLO = AR == 0 || AR == 2;           // the weight fragments carry a lo half
    constexpr int NT = 64 * WM * WN;                   // 4 waves (256 threads) or 8 waves (512 threads)
    constexpr bool M16 = CK == 32 && NTAPS == 9 && HD_MFMA16;   // 16 x 16 x 32 MFMA tiles (one instruction covers the whole 32-channel slice); the 1x1 kernels measured 5-12 % slower on it
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, PITCH = M16 ? 160 : 4 * CK + 16;
    constexpr int TMF = M16 ? 2 * TM : TM, TNF = M16 ? 2 * TN : TN;   // operand blocks per wave (16 or 32 rows / columns each)
    constexpr int KS = CK / 16;                        // k16 MFMA steps per tap
    constexpr int IPP = CK / 8;                        // 8-channel items per staged pixel
    constexpr int SLAB16 = BN * 64;                    // bytes of one k16-slab (global: contiguous; LDS: one ring slot)
    constexpr bool GL = conv_glds(WM, WN, CK, MODE, NTAPS);   // weights by LDS-DMA (per variant: conv_device.h)
    constexpr int GD = KS == 1 ? 2 : HD_GLDS_D;        // LDS-DMA: slabs requested ahead (16-channel slices: taps of 12 MFMAs are too short to cover one DMA's latency)
    constexpr int RING = (GL ? GD + 1 : 2) * KS;            // ring slots: the slabs (taps) being read, landing and -- with LDS-DMA -- in flight
    constexpr int NWH = SLAB16 / 16 / NT;              // 16-byte pieces per thread per k16-slab
    constexpr int NW = NWH * KS;                       // ... per weight unit (one tap's worth)
    constexpr bool XDB = NT == 512 && NTAPS == 9;      // two activation windows
    constexpr int NV = MODE == IN_NONE || MODE == IN_SOFTMAX32 ? 0 : MODE == IN_LAYERNORM ? 1 : MODE == IN_AFFINE_SILU ? 2 : 3;
    static_assert(MODE != IN_SOFTMAX32 || CK == 32, "the softmax loader works on 32-channel slices");
    static_assert(NWH >= 1 && NW <= 4 && SLAB16 % (16 * NT) == 0 && (NT == 256 || NT == 512), "bad tile");
    static_assert(NTAPS == 9 || NTAPS == 0, "taps are unrolled for 3x3 filters only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int npx = p.npx;
    const int npx4 = (npx + 3) & ~3;
    // LDS: [rowpix | rowb] (read by the epilogue) [weight ring] [pxsrc | pxb] [parameter table] [window(s)]; the epilogue's staging
    // area overlays everything from the ring on.  The ring sits at a compile-time offset.
    int* rowpix = reinterpret_cast<int*>(smem);
    int* rowb = rowpix + BM;
    char* Ws = reinterpret_cast<char*>(rowb + BM);     // [RING][BN][64], pieces XOR-swizzled
    int* pxsrc = reinterpret_cast<int*>(Ws + RING * SLAB16);
    int* pxb = pxsrc + npx4;
    const int n4 = p.pt_n4;                            // 16-byte entries per table vector; entry n4 is a write sink
    const int ptv = (n4 + 1) * 16, pt_stride = NV * ptv;
    char* ptab = reinterpret_cast<char*>(pxb + npx4);  // [2][NV][n4 + 1] float4
    char* Xs = ptab + 2 * pt_stride;                   // [1 or 2][npx + 1][PITCH]; row npx is a write sink
    const int xs_stride = XDB ? p.xs_stride : 0;

#ifdef HD_STAMPS
    const unsigned long long st_begin = HD_STAMP();
    unsigned long long st_bar = 0, st_issue = 0, st_arrive = 0, st_a = 0, st_b = 0, st_c = 0, st_d = 0;
#endif
    const TileCtx t = tile_decode<WN, BN>(p);
    const int tid = t.tid;
    // split-K (3x3 only): grid.y picks a run of K slices; this workgroup writes its raw sums to its own slab of the workspace
    int cb = 0;
    if constexpr (NTAPS == 9) {
        if (p.ksplit > 1) { cb = blockIdx.y * p.kchunks; p.out += (size_t)blockIdx.y * p.split_stride; }
    }

    // ---- parameter table: thread tid < n4 owns entry tid = (sample tid / (CK/4), channels 4 * (tid % (CK/4)))
    const int pt_dst = (tid < n4 ? tid : n4) * 16;
    int pt_src = 0;
    if constexpr (NV > 0) {
        const int e = tid < n4 ? tid : 0, bl = e / (CK / 4), q = e - bl * (CK / 4);
        const int b = t.b0 + bl < p.B ? t.b0 + bl : p.B - 1;
        pt_src = (MODE == IN_LAYERNORM ? 0 : b * p.in_bstride) + q * 4;
    }
    float4 pr0 = make_float4(0.f, 0.f, 0.f, 0.f), pr1 = pr0, pr2 = pr0;
    auto pt_load = [&](int c) {      // every lane loads (clamped entry): no branch around a memory operation
        if constexpr (NV > 0) {
            const int o = pt_src + (c + cb) * CK;
            pr0 = *reinterpret_cast<const float4*>((MODE == IN_LAYERNORM ? p.ln_g : p.inA) + o);
            if constexpr (NV > 1) pr1 = *reinterpret_cast<const float4*>(p.inB + o);
            if constexpr (NV > 2) pr2 = *reinterpret_cast<const float4*>(p.inE + o);
        }
    };
    auto pt_store = [&](int c) {
        if constexpr (NV > 0) {
            char* d = ptab + (c & 1) * pt_stride + pt_dst;
            *reinterpret_cast<float4*>(d) = pr0;
            if constexpr (NV > 1) *reinterpret_cast<float4*>(d + ptv) = pr1;
            if constexpr (NV > 2) *reinterpret_cast<float4*>(d + 2 * ptv) = pr2;
        }
    };
#ifdef HD_STAMPS
    st_a = HD_STAMP();                 // tile decoded
#endif
    pt_load(0);
    init_tables<BM, NT>(p, t, pxsrc, pxb, rowpix, rowb);
    pt_store(0);
#ifdef HD_STAMPS
    st_b = HD_STAMP();                 // tables written
#endif

    // operand fragment addresses.  A: staged pixel row of GEMM row (wm, tm, lane) + this lane's k half.  B: output channel n of the ring
    // slot, pieces swizzled: hi piece `half`, lo piece 2 + half, each XOR (n >> 2) & 3.
    // 16 x 16 x 32 tiles (M16): lane l holds row / column l & 15 and k chunk l >> 4 (8 channels) of the 32: A = bytes 16 * (l >> 4) of the pixel row's
    // hi (lo: + 64) half -- the 160-byte pitch keeps a 16-row block's reads conflict-free when its staged indices are consecutive modulo 16
    // (decode_row) --; B = k16-slab (l >> 5) of the tap, piece (l >> 4) & 1 (lo: + 2) of row n, XOR (n >> 2) & 2.
    int aoff[TMF], boff[TNF];
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < TMF; ++i) aoff[i] = row_px_offset_m(p, t, t.wm * 32 * TM + i * 16 + (t.lane & 15)) * PITCH + (t.lane >> 4) * 16;
#pragma unroll
        for (int j = 0; j < TNF; ++j) {
            const int n = t.wn * 32 * TN + j * 16 + (t.lane & 15), ch = t.lane >> 4;
            boff[j] = (ch >> 1) * SLAB16 + n * 64 + ((((n >> 2) & 2) ^ (ch & 1)) << 4);
        }
    } else {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) aoff[tm] = row_px_offset<TM>(p, t, tm) * PITCH + t.half * 16;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = t.wn * 32 * TN + tn * 32 + t.l31;
            boff[tn] = n * 64 + ((((n >> 2) & 3) ^ t.half) << 4);
        }
    }

    using AccT = std::conditional_t<M16, f32x4, f32x16>;
    AccT acc[TMF][TNF];
#pragma unroll
    for (int tm = 0; tm < TMF; ++tm)
#pragma unroll
        for (int tn = 0; tn < TNF; ++tn)
#pragma unroll
            for (int r = 0; r < (M16 ? 4 : 16); ++r) acc[tm][tn][r] = 0.f;

    const int ntaps = NTAPS ? NTAPS : p.KH * p.KW, nchunks_tot = p.Cin / CK;
    const int nchunks = (NTAPS == 9 && p.ksplit > 1) ? min(p.kchunks, nchunks_tot - cb) : nchunks_tot, nit = ntaps * nchunks;
    __syncthreads();                 // tables and the parameter table of slice 0 visible
#ifdef HD_STAMPS
    st_c = HD_STAMP();                 // first barrier passed
#endif

    // per-thread staging items: item i = tid + NT*j = (pixel i / IPP, channels 8*(i % IPP)..+7).  NT is a
    // multiple of IPP, so the channel offset q8 is the same for all of a thread's items and its pixels are
    // PXSTEP apart.  Source indices are clamped so every lane issues every load; padding pixels (src < 0)
    // are zeroed and items past the window go to the sink row when the LDS rows are written.
    constexpr int PXSTEP = NT / IPP;
    const int q8 = (tid % IPP) * 8, px0 = tid / IPP;
    int it_src[MAXI], it_dst[MAXI], it_pt[MAXI];
    bool it_pad[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int px = px0 + j * PXSTEP;
        const int pxc = px < npx ? px : npx - 1;
        const int s = pxsrc[pxc];
        it_pad[j] = s < 0;                                       // zero padding
        it_src[j] = s < 0 ? 0 : s;
        it_dst[j] = (px < npx ? px : npx) * PITCH + q8 * 2;
        it_pt[j] = ((MODE == IN_LAYERNORM ? 0 : pxb[pxc] - t.b0) * CK + q8) * 4;
    }
    float ln_mu[MAXI], ln_rs[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        ln_mu[j] = 0.f; ln_rs[j] = 1.f;
        if constexpr (MODE == IN_LAYERNORM) {
            const float2 st = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (size_t)it_src[j]);
            // The two statistics are copied out of the loaded register pair.  Used in place, hipcc broadcasts
            // rstd to both lanes of its packed multiplies straight from the pair's odd register (op_sel:[0,1]);
            // every build that did so produced exact zeros in the low lane for the last 16 lanes of a wave on
            // large grids (tests/test_gpu_kernels.py::test_large_grid_*), every build that did not was clean.
            // Root cause not established (open questions).
#ifdef HD_LN_INPLACE      // hazard bisect builds (tools/ln_zero_lane_bisect.sh): the statistics used in place, as hipcc allocates them
            ln_rs[j] = st.y; ln_mu[j] = st.x;
#if HD_LN_INPLACE > 1     // ... with every load retired before anything else happens
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#else
            asm volatile("v_mov_b32 %0, %1" : "=v"(ln_rs[j]) : "v"(st.y));
            asm volatile("v_mov_b32 %0, %1" : "=v"(ln_mu[j]) : "v"(st.x));
#endif
        }
    }

    // ---- weights.  Prefetch registers are named scalars (an indexed array here is left in scratch by hipcc); one SET holds a weight
    // unit = KS k16-slabs = NW 16-byte pieces per thread (piece k: k16-slab k / NWH of the unit, bytes (tid + (k % NWH) * NT) * 16 of it).
    // 3x3 kernels keep THREE sets in flight (unit U(it) travels in set it % 3; nine taps per slice make that a compile-time index): a
    // unit is requested three iterations before it is written to LDS.  The distance matters twice over: it covers the weights' own L2
    // latency, and -- vmcnt retiring in issue order -- the wait for a unit also waits for every activation load issued before that unit
    // was requested, so it is what gives the (HBM-latency) activation loads their three iterations of runway.
    constexpr int WD = NTAPS == 9 && !(NW == 4 && MAXI > 3) ? 3 : 1;   // (the largest-window 128 x 128 variant would lose its second workgroup per CU to the extra registers)
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0, v0 = w0, v1 = w0, v2 = w0, v3 = w0, u0 = w0, u1 = w0, u2 = w0, u3 = w0;
    int wdst[NWH];                                     // swizzled LDS offset of this thread's piece(s) inside a ring slot
#pragma unroll
    for (int kk = 0; kk < NWH; ++kk) {
        const int j = tid + kk * NT, row = j >> 2, pc = j & 3;
        wdst[kk] = row * 64 + ((pc ^ (M16 ? (row >> 2) & 2 : (row >> 2) & 3)) << 4);
    }
    // k16-slab (slice c, tap, step s) of this workgroup's K range; weights are stored [tap][Cin/16][CoutPad][64 B]; with per-sample
    // weights (w_bstride != 0, one sample per tile) the image of sample b0
    // LDS-DMA form: one wave-instruction moves 1 KiB = 16 rows; LDS takes it linearly (wave base + lane * 16), so the swizzle is applied to
    // the SOURCE: lane l fills slot l & 3 of row l >> 2, which must hold piece (l & 3) ^ ((row >> 2) & 3) = (l & 3) ^ ((l >> 4) & 3).
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gl_lane = ((t.lane >> 2) << 6) + (((t.lane & 3) ^ ((t.lane >> 4) & (M16 ? 2 : 3))) << 4);
    const char* wbase = reinterpret_cast<const char*>(p.wsplit) + (size_t)t.b0 * p.w_bstride + (size_t)t.n0 * 64 + (GL ? gl_lane : tid * 16);
    const int nch16 = nchunks_tot * KS;
    auto k16_src = [&](int c, int tap, int s) { return wbase + (size_t)(tap * nch16 + (c + cb) * KS + s) * p.CoutPad * 64; };
    // slab (c, tap) -> ring slots slot0 .. slot0 + KS - 1, NW wave-instructions per wave, nothing waits here
    auto w_glds = [&](int c, int tap, int slot0) {
        if (ABL(256)) return;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int h = k / NWH, i = (k % NWH) * (NT / 64) + wv;          // k16-slab of the slab, 1 KiB piece inside it
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(k16_src(c, tap, h) + i * 1024),
                                             (__attribute__((address_space(3))) void*)(Ws + (slot0 + h) * SLAB16 + i * 1024), 16, 0, 0);
        }
    };
#define HD_WLOAD(S, k, src)  if constexpr (NW > k) { if (!ABL(256)) S##k = *reinterpret_cast<const uint4*>((src) + ((k) % NWH) * NT * 16); }
#define HD_WSTORE(S, k, slot) if constexpr (NW > k) { if (!ABL(4)) *reinterpret_cast<uint4*>(Ws + (slot) * SLAB16 + wdst[(k) % NWH]) = S##k; }
    // piece k of set S <- k16-slab h = k / NWH of the unit, whose source is src[h] (nullptr: that k16-slab does not exist -- past the end
    // of the K range; decided at compile time by the callers, so no branch survives)
    auto w_load = [&](auto set, auto have, const char* s0, const char* s1) {
        constexpr int S = decltype(set)::value;
        constexpr int H = decltype(have)::value;      // bit h: k16-slab h of the unit exists
#define HD_WL(SS, k) if constexpr ((H >> ((k) / NWH)) & 1) { HD_WLOAD(SS, k, ((k) / NWH) ? s1 : s0) }
        if constexpr (S == 0) { HD_WL(w, 0) HD_WL(w, 1) HD_WL(w, 2) HD_WL(w, 3) }
        else if constexpr (S == 1) { HD_WL(v, 0) HD_WL(v, 1) HD_WL(v, 2) HD_WL(v, 3) }
        else { HD_WL(u, 0) HD_WL(u, 1) HD_WL(u, 2) HD_WL(u, 3) }
#undef HD_WL
    };
    auto w_store = [&](auto set, auto have, int slot0, int slot1) {
        constexpr int S = decltype(set)::value;
        constexpr int H = decltype(have)::value;
#define HD_WS(SS, k) if constexpr ((H >> ((k) / NWH)) & 1) { HD_WSTORE(SS, k, ((k) / NWH) ? slot1 : slot0) }
        if constexpr (S == 0) { HD_WS(w, 0) HD_WS(w, 1) HD_WS(w, 2) HD_WS(w, 3) }
        else if constexpr (S == 1) { HD_WS(v, 0) HD_WS(v, 1) HD_WS(v, 2) HD_WS(v, 3) }
        else { HD_WS(u, 0) HD_WS(u, 1) HD_WS(u, 2) HD_WS(u, 3) }
#undef HD_WS
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using HAll = std::integral_constant<int, (1 << KS) - 1>;

    // raw activation prefetch registers.  The any-filter kernels with small windows keep TWO sets (slices c+1 and c+2 in flight): with one tap
    // per slice a set requested during slice c is consumed a single MFMA cluster later, far inside the HBM latency (the 1x1 shortcut at 64 x 64
    // ran at 57 % of its HBM share, every slice an exposed round trip).  The set index is a compile-time constant at every use.
    constexpr int XD = NTAPS == 0 && MAXI <= 4 && HD_CONV_XD2 ? 2 : 1;
    float4 xr[XD][MAXI][2];
    using X0 = std::integral_constant<int, 0>;
    using X1 = std::integral_constant<int, XD - 1>;
    auto x_load_s = [&](auto set, int j, const float* src, int Csrc) {       // src already points at (slice, q8)
        if (ABL(32)) return;
        constexpr int Q = decltype(set)::value;
        const float* g = src + (size_t)it_src[j] * Csrc;
        xr[Q][j][0] = *reinterpret_cast<const float4*>(g);
        xr[Q][j][1] = *reinterpret_cast<const float4*>(g + 4);
    };
    auto x_load = [&](int j, const float* src, int Csrc) { x_load_s(X0{}, j, src, Csrc); };
    // transform + bf16 hi/lo split + LDS write of item j (raw values in xr, slice c) into window xdst
    auto x_stage_s = [&](auto set, int j, int c, char* xdst) {
        if (ABL(64)) return;
        constexpr int Q = decltype(set)::value;
        const char* pt = ptab + (c & 1) * pt_stride + it_pt[j];
        float4 v0 = xform4<MODE>(xr[Q][j][0], pt, ptv, ln_mu[j], ln_rs[j]);
        float4 v1 = xform4<MODE>(xr[Q][j][1], pt + 16, ptv, ln_mu[j], ln_rs[j]);
        if constexpr (MODE == IN_SOFTMAX32) softmax32(v0, v1);
        uint4 hi, lo;
        if constexpr (F16) { hi = half8(v0, v1); lo = hi; }
        else split8(v0, v1, hi, lo);
        if (it_pad[j]) { hi = make_uint4(0, 0, 0, 0); lo = hi; }   // zero padding is applied AFTER the transform
        char* d = xdst + it_dst[j];
        *reinterpret_cast<uint4*>(d) = hi;
        if constexpr (ALO) *reinterpret_cast<uint4*>(d + 2 * CK) = lo;      // (PLAIN / F16W2: the lo half is dead code)
    };
    auto x_stage = [&](int j, int c, char* xdst) { x_stage_s(X0{}, j, c, xdst); };
    auto slice_src = [&](int c, const float*& src, int& Csrc) {  // channel-concatenated input: two tensors
        const int cc = (c + cb) * CK;
        if (cc < p.C0) { src = p.in0 + cc + q8; Csrc = p.C0; } else { src = p.in1 + (cc - p.C0) + q8; Csrc = p.C1; }
    };

    // ---- operand fragments: A from the window (pixel rows; k-step s of the slice at byte s*32 of the hi / lo halves), B from ring slot
    struct FragA { bf16x8 h[TMF], l[TMF]; };
    struct FragB { bf16x8 h[TNF], l[TNF]; };
    auto ld_a = [&](FragA& f, const char* Xc, int off) {          // off = (tap offset in pixels) * PITCH (+ s * 32 for the 16-channel steps of the 32 x 32 tiling)
#pragma unroll
        for (int tm = 0; tm < TMF; ++tm) {
            const char* a = Xc + aoff[tm] + off;
            f.h[tm] = *reinterpret_cast<const bf16x8*>(a);
            if constexpr (ALO) f.l[tm] = *reinterpret_cast<const bf16x8*>(a + 2 * CK);
        }
    };
    auto ld_b = [&](FragB& f, int slot) {
#pragma unroll
        for (int tn = 0; tn < TNF; ++tn) {
            const char* b = Ws + slot * SLAB16;
            f.h[tn] = *reinterpret_cast<const bf16x8*>(b + boff[tn]);
            if constexpr (BLO) f.l[tn] = *reinterpret_cast<const bf16x8*>(b + (boff[tn] ^ 32));
        }
    };
    auto mma = [](const bf16x8& a, const bf16x8& b, const AccT& c) {
        if constexpr (F16) {                           // the same 16 bytes per fragment, read as fp16
            const f16x8 ah = __builtin_bit_cast(f16x8, a), bh = __builtin_bit_cast(f16x8, b);
            if constexpr (M16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
        } else {
            if constexpr (M16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        }
    };
    auto mm = [&](const FragA& a, const FragB& b) {
        if (ABL(8)) return;
#pragma unroll
        for (int tm = 0; tm < TMF; ++tm)
#pragma unroll
            for (int tn = 0; tn < TNF; ++tn) {
                if constexpr (ALO) acc[tm][tn] = mma(a.l[tm], b.h[tn], acc[tm][tn]);
                if constexpr (BLO) acc[tm][tn] = mma(a.h[tm], b.l[tn], acc[tm][tn]);
                acc[tm][tn] = mma(a.h[tm], b.h[tn], acc[tm][tn]);
            }
    };
    // a whole tap, fragments read in the step that uses them
    auto mfma_cluster = [&](const char* Xc, int slot0, int tapoff) {
#pragma unroll
        for (int s = 0; s < (M16 ? 1 : KS); ++s) {
            FragA a; FragB b;
            ld_a(a, Xc, tapoff + s * 32);
            ld_b(b, slot0 + s);
            mm(a, b);
        }
    };

    // schedule of the next slice's items over the taps of a 3x3 filter: requested during the first seven
    // taps; with two windows, staged three taps after the request (about three MFMA clusters of cover)
    constexpr int IPT = (MAXI + 6) / 7;
    constexpr int XLAT = 3;

#ifdef HD_STAMPS
    st_d = HD_STAMP();                 // per-thread items set up
#endif
    int par = 0;                       // register path: iteration parity; slab `it` sits in ring slots KS * par ..
    if constexpr (NTAPS == 9) {
        // ---- prologue: slice 0 staged; LDS-DMA: slabs 0 and 1 on their way to ring slots 0 .. 2 KS - 1; register path: slab 0 staged, slabs 1 .. WD in flight
        const float* src; int Csrc;
        slice_src(0, src, Csrc);
#pragma unroll
        for (int j = 0; j < MAXI; ++j) x_load(j, src, Csrc);
        if constexpr (GL) { w_glds(0, 0, 0); if constexpr (GD == 2) w_glds(0, 1, KS); }
        else w_load(S0{}, HAll{}, k16_src(0, 0, 0), k16_src(0, 0, KS - 1));
#ifdef HD_STAMPS
        st_issue = HD_STAMP();
        __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0) (gfx9 encoding: lgkmcnt / expcnt untouched): how long do the first loads take to arrive?
        st_arrive = HD_STAMP();
#endif
#pragma unroll
        for (int j = 0; j < MAXI; ++j) x_stage(j, 0, Xs);
        if constexpr (GL) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of slabs 0 and 1 have landed (the first tap's barrier publishes them)
        } else {
            w_store(S0{}, HAll{}, 0, 1);
            if constexpr (WD == 3) { w_load(S1{}, HAll{}, k16_src(0, 1, 0), k16_src(0, 1, KS - 1)); w_load(S2{}, HAll{}, k16_src(0, 2, 0), k16_src(0, 2, KS - 1)); w_load(S0{}, HAll{}, k16_src(0, 3, 0), k16_src(0, 3, KS - 1)); }
            else w_load(S0{}, HAll{}, k16_src(0, 1, 0), k16_src(0, 1, KS - 1));
        }
    } else {
        // ---- prologue (any filter): slice 0 and slab 0 staged, slab 1 in flight; with two register sets slice 1 is on its way too
        const float* src; int Csrc;
        slice_src(0, src, Csrc);
#pragma unroll
        for (int j = 0; j < MAXI; ++j) x_load(j, src, Csrc);
        w_load(S0{}, HAll{}, k16_src(0, 0, 0), KS > 1 ? k16_src(0, 0, 1) : nullptr);
        if constexpr (XD == 2) {
            const float* s1; int C1;
            slice_src(nchunks > 1 ? 1 : 0, s1, C1);          // (clamped: a redundant load when there is one slice only)
#pragma unroll
            for (int j = 0; j < MAXI; ++j) x_load_s(X1{}, j, s1, C1);
        }
#ifdef HD_STAMPS
        st_issue = HD_STAMP();
        __builtin_amdgcn_s_waitcnt(0x0070);
        st_arrive = HD_STAMP();
#endif
#pragma unroll
        for (int j = 0; j < MAXI; ++j) x_stage(j, 0, Xs);
        w_store(S0{}, HAll{}, 0, 1);
        if (nit > 1) {
            if (ntaps > 1) w_load(S0{}, HAll{}, k16_src(0, 1, 0), KS > 1 ? k16_src(0, 1, 1) : nullptr);
            else w_load(S0{}, HAll{}, k16_src(1, 0, 0), KS > 1 ? k16_src(1, 0, 1) : nullptr);
        }
    }

#ifdef HD_STAMPS
    const unsigned long long st_loop = HD_STAMP();
#endif
    if constexpr (NTAPS == 9) {
        // One tap of the main loop, tap index known at compile time.  NEXT: a next slice exists (slices
        // 0 .. nchunks-2), so the tap stores/requests unconditionally; the last slice only drains its own slabs.
        auto tap_body = [&](auto tapc, auto has_next, int c, const char* Xc, char* Xn, const float* nsrc, int nCsrc) {
            constexpr int tap = decltype(tapc)::value;
            constexpr bool NEXT = decltype(has_next)::value;
#ifdef HD_STAMPS
            const unsigned long long b0 = HD_STAMP();
#endif
            // slab `it` and the window of slice c are visible; nobody still reads slab it-1.  LDS-DMA path: a raw barrier -- __syncthreads() would
            // drain the weight DMAs still in flight (hipcc puts vmcnt(0) in front of it); every wave waited for its own pieces of slab `it` at the
            // end of the previous tap, and the "memory" clobber keeps hipcc's LDS accesses on their side of the barrier.
            if constexpr (GL) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else __syncthreads();
#ifdef HD_STAMPS
            st_bar += HD_STAMP() - b0;
#endif
            if constexpr (GL && XDB && NEXT) {      // (before this tap's DMAs are issued: hipcc waits vmcnt(0) where these items' loads are used)
                if constexpr (tap == 2) pt_store(c + 1);
#pragma unroll
                for (int j = 0; j < MAXI; ++j)
                    if (j / IPT + XLAT == tap || (tap == 8 && j / IPT + XLAT > 8)) x_stage(j, c + 1, Xn);
            }
            auto x_req = [&]() {
                if constexpr (NEXT) {
                    if constexpr (tap == 0) pt_load(c + 1);
#pragma unroll
                    for (int j = 0; j < MAXI; ++j)
                        if (j / IPT == tap) x_load(j, nsrc, nCsrc);
                }
            };
            if constexpr (!GL || GD == 2) x_req();
            if constexpr (GL) {
                // slab it+GD -> the ring slots slab it-1 was read from.  The fence pins its place among this tap's register loads so that the count
                // at the end of the tap is exact (vmcnt retires in issue order): with two slabs ahead the DMAs go last, with one they go first.
                __builtin_amdgcn_sched_barrier(0);
                constexpr int t2 = tap + GD;
                const int slot = GD == 2 ? (t2 % 3) * KS : KS * (par ^ 1);
                if constexpr (t2 < 9) w_glds(c, t2, slot);
                else if constexpr (NEXT) w_glds(c + 1, t2 - 9, slot);
                if constexpr (GD == 1) { __builtin_amdgcn_sched_barrier(0); x_req(); }
            } else {
                // slab it+1 (set (tap+1) % WD) -> the other half of the ring, then slab it+1+WD is requested into the set just freed
                using SO = std::integral_constant<int, (tap + 1) % WD>;
                if constexpr (NEXT || tap + 1 < 9) w_store(SO{}, HAll{}, KS * (par ^ 1), KS * (par ^ 1) + 1);
                constexpr int tl = tap + 1 + WD;
                if constexpr (tl < 9) w_load(SO{}, HAll{}, k16_src(c, tl, 0), k16_src(c, tl, KS - 1));
                else if constexpr (NEXT) w_load(SO{}, HAll{}, k16_src(c + 1, tl - 9, 0), k16_src(c + 1, tl - 9, KS - 1));
            }
            if constexpr (!GL && XDB && NEXT) {
                if constexpr (tap == 2) pt_store(c + 1);
#pragma unroll
                for (int j = 0; j < MAXI; ++j)
                    if (j / IPT + XLAT == tap || (tap == 8 && j / IPT + XLAT > 8)) x_stage(j, c + 1, Xn);
            }
            mfma_cluster(Xc, GL && GD == 2 ? (tap % 3) * KS : KS * par, ((tap / 3) * p.LW + tap % 3) * PITCH);
            if constexpr (GL) {
                // slab it+1 (requested during tap it-1) must have landed before the next barrier.  Younger than it in the queue: this tap's register
                // loads (two per requested item, the parameter-table vectors at tap 0) and this tap's DMAs; everything older retires first.
                constexpr int nx = NEXT ? 2 * ((tap + 1) * IPT < MAXI ? IPT : (tap * IPT < MAXI ? MAXI - tap * IPT : 0)) + (tap == 0 ? NV : 0) : 0;
                constexpr int ng = GD == 2 && (NEXT || tap + 2 < 9) ? NW : 0;
                if constexpr (NEXT || tap + 1 < 9) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nx + ng) : "memory");
            }
            par ^= 1;
        };
#define HD_NINE_TAPS(NEXT_T, ...)                                                                          \
        tap_body(std::integral_constant<int, 0>{}, NEXT_T{}, __VA_ARGS__); tap_body(std::integral_constant<int, 1>{}, NEXT_T{}, __VA_ARGS__); \
        tap_body(std::integral_constant<int, 2>{}, NEXT_T{}, __VA_ARGS__); tap_body(std::integral_constant<int, 3>{}, NEXT_T{}, __VA_ARGS__); \
        tap_body(std::integral_constant<int, 4>{}, NEXT_T{}, __VA_ARGS__); tap_body(std::integral_constant<int, 5>{}, NEXT_T{}, __VA_ARGS__); \
        tap_body(std::integral_constant<int, 6>{}, NEXT_T{}, __VA_ARGS__); tap_body(std::integral_constant<int, 7>{}, NEXT_T{}, __VA_ARGS__); \
        tap_body(std::integral_constant<int, 8>{}, NEXT_T{}, __VA_ARGS__);
        for (int c = 0; c + 1 < nchunks; ++c) {
            const float* nsrc; int nCsrc;
            slice_src(c + 1, nsrc, nCsrc);
            const char* Xc = Xs + (c & 1) * xs_stride;
            char* Xn = Xs + ((c + 1) & 1) * xs_stride;
            HD_NINE_TAPS(std::true_type, c, Xc, Xn, nsrc, nCsrc)
            if constexpr (!XDB) {
                pt_store(c + 1);
                __syncthreads();         // single window: every wave has finished the slice
#pragma unroll
                for (int j = 0; j < MAXI; ++j) x_stage(j, c + 1, Xs);
            }
        }
        {
            const int c = nchunks - 1;
            const char* Xc = Xs + (c & 1) * xs_stride;
            HD_NINE_TAPS(std::false_type, c, Xc, nullptr, nullptr, 0)
        }
#undef HD_NINE_TAPS
    } else {
        // ---- any filter shape: taps in a loop; tap 0 (which requests the whole next slice) and the last slice
        // are peeled.  Slab indices past the end are clamped instead of branched around (a redundant load / a
        // store to the idle slots).  Slab L (all its k16 steps) lives in ring slots KS * (L & 1) ..
        int it = 0;
        auto w_next = [&](int k) {
            const int kk = k < nit ? k : nit - 1; const int c = kk / ntaps, tp = kk - c * ntaps;
            w_load(S0{}, HAll{}, k16_src(c, tp, 0), KS > 1 ? k16_src(c, tp, 1) : nullptr);
        };
        // PAR: parity of c = the register set that is free during slice c (it held slice c, staged at the end of slice c-1; with one set: the only one).
        // Two sets: slice c+2 is requested into set PAR (clamped to the last slice past the end: a redundant load instead of a branch) and slice
        // c+1 is staged from the other set; one set: slice c+1 is requested and staged within slice c.
        auto slice = [&](int c, auto has_next, auto parc) {
            constexpr bool NEXT = decltype(has_next)::value;
            using QL = std::integral_constant<int, XD == 2 ? decltype(parc)::value : 0>;          // set to load into
            using QS = std::integral_constant<int, XD == 2 ? 1 - decltype(parc)::value : 0>;      // set to stage from
            const float* nsrc = nullptr; int nCsrc = 0;
            if constexpr (NEXT) slice_src(XD == 2 ? min(c + 2, nchunks - 1) : c + 1, nsrc, nCsrc);
            {
                __syncthreads();         // slab `it` and the window of slice c are visible; nobody still reads slab it-1
                w_store(S0{}, HAll{}, KS * (par ^ 1), KS * (par ^ 1) + 1);
                w_next(it + 2);
                if constexpr (NEXT) {
                    pt_load(c + 1);
#pragma unroll
                    for (int j = 0; j < MAXI; ++j) x_load_s(QL{}, j, nsrc, nCsrc);
                }
                mfma_cluster(Xs, KS * par, 0);
                par ^= 1; ++it;
            }
            int ky = 0, kx = 1;
            if (kx == p.KW) { kx = 0; ky = 1; }
            for (int tap = 1; tap < ntaps; ++tap, ++it) {
                __syncthreads();
                w_store(S0{}, HAll{}, KS * (par ^ 1), KS * (par ^ 1) + 1);
                w_next(it + 2);
                mfma_cluster(Xs, KS * par, (ky * p.LW + kx) * PITCH);
                par ^= 1;
                if (++kx == p.KW) { kx = 0; ++ky; }
            }
            if constexpr (NEXT) {
                pt_store(c + 1);
                __syncthreads();         // single window: every wave has finished the slice
#pragma unroll
                for (int j = 0; j < MAXI; ++j) x_stage_s(QS{}, j, c + 1, Xs);
            }
        };
        {
            int c = 0;
            for (; c + 2 < nchunks; c += 2) { slice(c, std::true_type{}, X0{}); slice(c + 1, std::true_type{}, std::integral_constant<int, 1>{}); }
            if (c + 1 < nchunks) { slice(c, std::true_type{}, X0{}); slice(c + 1, std::false_type{}, std::integral_constant<int, 1>{}); }
            else slice(c, std::false_type{}, X0{});
        }
    }
    (void)w1; (void)w2; (void)w3; (void)v0; (void)v1; (void)v2; (void)v3; (void)u0; (void)u1; (void)u2; (void)u3; (void)pr1; (void)pr2; (void)wv; (void)wdst;
#undef HD_WLOAD
#undef HD_WSTORE
#ifdef HD_STAMPS
    const unsigned long long st_epi = HD_STAMP();
#endif
    // epilogue modes an instantiation can be asked for (conv_host.hip: conv_ep_mask): the unrolled-tap kernels serve the GroupNorm'd / FiLM'd /
    // residual 3x3 layers, the any-filter kernels the 1x1 projections with their LayerNorm and GroupNorm-apply tails; what a family never
    // sees is compiled out of it (the 8-wave kernel: 186.0 -> 184.9 ms per hicedrn64 step, A/B on one box)
    constexpr int EPMASK = NTAPS == 9 ? (EP_FILM_SILU | EP_ADD_SILU | EP_RES | EP_FILM_SILU_BWD) : (EP_RES | EP_RES_AFFINE_SILU | EP_LN_RES | EP_LN_STATS);
    conv_epilogue<BM, BN, TM, TN, NT, FBWD, EPMASK>(p, t, acc, rowpix, rowb, reinterpret_cast<float*>(Ws));
#ifdef HD_STAMPS
    if (tid == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* g = p.stamps + 16 * blockIdx.x;
        g[0] = st_begin; g[1] = st_loop; g[2] = st_epi; g[3] = HD_STAMP(); g[4] = st_bar; g[5] = ((unsigned long long)xcc << 32) | hwid; g[6] = st_issue - st_begin; g[7] = st_arrive - st_issue; g[8] = st_a - st_begin; g[9] = st_b - st_a; g[10] = st_c - st_b; g[11] = st_d - st_c;
    }
#endif
}

template <int WM, int WN, int TM, int TN, int CK, int MAXI, int MODE, int NTAPS>
// (16-channel slices with the 256 x 64 tile: ~40 KB of LDS, so a third workgroup fits a CU if the registers allow -- at most 168.  The same cap on
// the 128 x 128 variants with 16-channel slices everywhere measured a loss: 14.1 vs 13.45 ms per unet64 step.)
// Only the plain loader fits 168 registers without scratch (the GroupNorm-apply loaders need 196: capped, they spill 12 registers -- tests/test_isa_guards.py).
__global__ __launch_bounds__(64 * WM * WN, (conv_cap3<WM, WN, CK, MODE, NTAPS>() ? 3 : (NTAPS == 0 && MAXI > 4) ? 1 : 2)) void conv_igemm_bf16x3_kernel(ConvKArgs p) {
    conv_igemm_bf16x3_body<WM, WN, TM, TN, CK, MAXI, MODE, NTAPS, 0>(p);
}
// two fp16 products per multiply, xh (wh + wl): the 3x3 kernels only (the any-filter kernels are not MFMA-bound)
template <int WM, int WN, int TM, int TN, int CK, int MAXI, int MODE, int NTAPS>
__global__ __launch_bounds__(64 * WM * WN, (conv_cap3<WM, WN, CK, MODE, NTAPS>() ? 3 : 2)) void conv_igemm_f16w2_kernel(ConvKArgs p) {
    static_assert(NTAPS == 9, "3x3 kernels only");
    conv_igemm_bf16x3_body<WM, WN, TM, TN, CK, MAXI, MODE, NTAPS, 2>(p);
}
// one fp16 product per multiply, xh wh
template <int WM, int WN, int TM, int TN, int CK, int MAXI, int MODE, int NTAPS>
__global__ __launch_bounds__(64 * WM * WN, (conv_cap3<WM, WN, CK, MODE, NTAPS>() ? 3 : 2)) void conv_igemm_f16w1_kernel(ConvKArgs p) {
    static_assert(NTAPS == 9, "3x3 kernels only");
    conv_igemm_bf16x3_body<WM, WN, TM, TN, CK, MAXI, MODE, NTAPS, 3>(p);
}
template <int WM, int WN, int TM, int TN, int CK, int MAXI, int MODE, int NTAPS>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_bf16_kernel(ConvKArgs p) {
    conv_igemm_bf16x3_body<WM, WN, TM, TN, CK, MAXI, MODE, NTAPS, 1>(p);
}
// training only: the same bodies with the EP_FILM_SILU_BWD epilogue mode compiled in (PLAIN: the bf16 option's one-product form)
template <int WM, int WN, int TM, int TN, int CK, int MAXI, int MODE, int NTAPS, bool PLAIN>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_bf16x3_fbwd_kernel(ConvKArgs p) {
    conv_igemm_bf16x3_body<WM, WN, TM, TN, CK, MAXI, MODE, NTAPS, PLAIN ? 1 : 0, true>(p);
}

template <typename K>
static int launch_one(K kernel, const char* name, ConvLaunch& L, hipStream_t st, int nthreads = 256) {
    static std::set<const void*> raised;   // every instantiation has the same pointer TYPE: key by address
    if (!raised.count(reinterpret_cast<const void*>(kernel))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hd_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); return -3;
        }
        raised.insert(reinterpret_cast<const void*>(kernel));
    }
    const ConvKArgs& k = L.k;
    const int mtiles = ((k.B + k.TB - 1) / k.TB) * k.tiles_y * k.tiles_x;
    dim3 grid((unsigned)(mtiles * k.ntiles_n), (unsigned)(k.ksplit > 1 ? k.ksplit : 1));
    conv_prof_begin(L, name, st);
    hipLaunchKernelGGL(kernel, grid, dim3(nthreads), L.lds, st, L.k);
    conv_prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("conv launch: ") + hipGetErrorString(e)); return -3; }
    return 0;
}

// kernel-side loader mode of a launch
static inline int conv_kernel_mode(const ConvLaunch& L) {
    if (L.k.in_mode == IN_AFFINE_SILU) return L.k.inE ? IN_AFFINE_SILU_E : IN_AFFINE_SILU;
    return L.k.in_mode;   // IN_NONE, IN_LAYERNORM, IN_SOFTMAX32
}

// Largest window (in staged 8-channel items per thread) a launch may use; conv_host.hip plans the tile inside
// it and the dispatch below picks the smallest instantiated variant that holds the window.
int conv_bf16x3_max_items(int cfg, int ck, bool taps9, bool layernorm);

