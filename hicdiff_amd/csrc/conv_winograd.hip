// Winograd F(2x2, 3x3) form of the stride-1 3x3 convolutions, split-bf16 x3 products on v_mfma_f32_32x32x16_bf16.
//
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray; d = 4x4 input patch, Y = 2x2 outputs)
//
// 16 element-wise positions, each a channel contraction = a GEMM over Cin: 16 x (tiles x Cin) x (Cin x Cout)
// instead of 9 x (4 tiles) x ... -> 2.25 x fewer MFMAs than the implicit-GEMM kernel.  The filter transform is done
// once at weight-pack time (fp64, then split into bf16 hi/lo); the input transform (fp32 adds) happens while the
// window is staged, the output transform in the epilogue, so the numerics are those of the direct split-bf16 kernel
// (tests/studies/winograd_error_study.py: 1.8e-5 per forward, 8e-5 on a 50-step chain, direct: 1.6e-5 / 8e-5).
//
// Workgroup (8 waves, one per CU): one sample's 16 x 16 output pixels = 8 x 8 Winograd tiles (M = 64 rows per
// position) x 64 output channels, all 16 positions: 64 x 64 x 16 fp32 accumulators = half of the CU's register file.
// Wave w owns position row i = (w >> 1) & 3 ... see `wi`: positions (i, 0..3), and the 32-channel half nb.
//
// K loop, 16 input channels per stage, two barriers, the two wave groups (waves 0-3 / 4-7: the two waves of every
// SIMD) in opposite roles -- one on the matrix pipe, its SIMD partner on VALU + LDS:
//     B1   | group 0: T, channels 0-7 of window s+1 --B^T d B, split--> V[(s+1)&1] | group 1: A (its window plane), M(s)
//     Bmid | group 0: A (its window plane), M(s)                                      | group 1: T, channels 8-15
//   T: thread = (tile, channel pair): 16 window reads, 32 + 32 adds, 16 hi/lo splits, 32 LDS writes.
//   A: raw slice (registers, requested a stage earlier) --GroupNorm-apply + SiLU--> window plane; next request.
//   M: 4 positions x (2 tile blocks x 3 products) MFMAs; the next stage's weight fragments are requested as each
//      position's registers fall free.
//   LDS: V[2][16 positions][64 tiles][16 ch hi | 16 ch lo] (128 KiB, 64-byte rows, 16-byte chunks XOR-swizzled by
//   (tile >> 2) & 3 so every ds_read_b128 group hits 16 different bank quads) + the fp32 window, one plane of 8
//   channels per wave group, pixels stored [row][x parity][x / 2] so the 8 tiles of a tile row read 256 contiguous
//   bytes (20 KiB; a pixel's GroupNorm-apply + SiLU is evaluated once, not once per patch).  Each plane is read (T)
//   and refilled (A) by its own group in different half-stages, so one copy suffices.
//   Weights never touch LDS: the packed image is in MFMA-fragment order, every fragment belongs to exactly one wave,
//   which streams it L2 -> registers one stage ahead.
// Epilogue: the j half of the output transform in registers, the i half after one exchange through LDS
// (S[i][b][tile][channel], overlays V), then bias / FiLM / residual / GroupNorm partial sums and 256-byte row stores.
#include "conv_device.h"

#include <set>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Diagnostic switches (HICDIFF_ABLATE, make DIAG=1): timing experiments only, compiled out of the product build.
// 1 no output stores, 2 no window staging (A part), 4 no weight stream, 8 no MFMA, 16 no input transform (T part), 32 no exchange
#ifdef HD_DIAG
#define ABL(bit) (p.ablate & (bit))
#else
#define ABL(bit) false
#endif

#ifdef HD_DIAG
__device__ unsigned long long g_wino_stamps[2][8];   // [group][T part, M part, wait at B1, wait at Bmid, total loop, epilogue 1, epilogue 2, -]: cycles of workgroup 5, waves 0 / 4
#define STAMP() __builtin_readcyclecounter()
#endif

namespace {

constexpr int WIN = 18;                    // window side: 16 output pixels + halo
constexpr int NPX = WIN * WIN;             // 324
constexpr int PLANE = WIN * 2 * 9 * 32;    // bytes of one 8-channel plane: [18 rows][2 x parities][9][8 ch]
constexpr int VBUF = 16 * 64 * 64;         // bytes of one V buffer
constexpr int W2_OFF = 2 * VBUF;
constexpr int MAXW = 3;                    // window items (pixel, 4 channels) per thread and plane: 648 / 256
constexpr int TAB_OFF = W2_OFF + 2 * PLANE + 16;   // per-thread item tables: source pixel (int), window offset (ushort)
constexpr int LDS_BYTES = TAB_OFF + MAXW * 512 * 6;
enum { WMODE_NONE = 0, WMODE_AFFINE = 1, WMODE_AFFINE_E = 2 };

__device__ __forceinline__ unsigned pack_hi_lo(float x, float y, unsigned& lo) {
    const bf16x2 h = {(__bf16)x, (__bf16)y};
    const float hx = (float)h[0], hy = (float)h[1];
    const bf16x2 l = {(__bf16)(x - hx), (__bf16)(y - hy)};
    lo = __builtin_bit_cast(unsigned, l);
    return __builtin_bit_cast(unsigned, h);
}

__device__ __forceinline__ int win_off(int wy, int wx) { return ((wy * 2 + (wx & 1)) * 9 + (wx >> 1)) * 32; }

template <int MODE, int UP>
__global__ __launch_bounds__(512, 2) void conv_winograd_bf16x3_kernel(ConvKArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Vs = smem;
    char* const W2 = smem + W2_OFF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                         // role group = window plane (channels 8 grp .. 8 grp + 7 of a stage)
    const int l31 = lane & 31, half = lane >> 5;
    const int wi = wave & 3, nb = wave >> 2;            // position row and 32-channel half of this wave's accumulators

    // workgroup -> (sample, block row, block column, 64-channel tile); same XCD remap as the implicit-GEMM kernel
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, per = nwg >> 3;
        if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
    }
    const int ntn = p.ntiles_n;
    const int nt = bid % ntn;
    int mt = bid / ntn;
    const int bx = mt % p.tiles_x; mt /= p.tiles_x;
    const int by = mt % p.tiles_y;
    const int bs = mt / p.tiles_y;
    const int n0 = nt * 64, Y0 = by * 16, X0 = bx * 16;
    const int ns = p.Cin >> 4;
    const int u = tid & 255;

    // ---- A part: this thread's items of its group's window plane (pixel e >> 1, channel quad e & 1 = u & 1).  The per-item source
    // pixel (-1: zero padding / no item) and window offset live in a small LDS table: registers are the scarce resource here, and
    // a spilled loop invariant would be reloaded through vmcnt, draining the prefetches in flight.
    const int qd = u & 1;
    int* const srcT = reinterpret_cast<int*>(smem + TAB_OFF);                         // [MAXW][512]
    unsigned short* const dstT = reinterpret_cast<unsigned short*>(smem + TAB_OFF + MAXW * 512 * 4);   // [MAXW][512]
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
        const int e = u + 256 * k;
        const int px = e >> 1;
        const int wy = px / WIN, wx = px - wy * WIN;
        const int y = Y0 + wy - 1, x = X0 + wx - 1;
        const bool in = px < NPX && y >= 0 && y < p.H && x >= 0 && x < p.W;
        int sp = -1;
        if (in) sp = bs * p.IH * p.IW + (UP ? ((y >> 1) * p.IW + (x >> 1)) : (y * p.IW + x));
        srcT[k * 512 + tid] = sp;
        dstT[k * 512 + tid] = (unsigned short)(px < NPX ? grp * PLANE + win_off(wy, wx) + qd * 16 : 2 * PLANE);   // past the window: a 16-byte write sink
    }
    float4 xr[MAXW];
#pragma unroll
    for (int k = 0; k < MAXW; ++k) xr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // GroupNorm-apply parameters of the plane's 8 channels: wave-uniform addresses -> scalar loads into SGPRs (no VGPR is held
    // across the T part); a lane picks its quad's four at store time
    float sA[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sB[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sE[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto a_request = [&](int s) {
        if (ABL(2)) return;
        if (s >= ns) s = ns - 1;                        // past the end: a redundant (valid) load whose result is never used
        const int cc = (s << 4) + grp * 8 + qd * 4;
        const float* base; int Cs;
        if (cc < p.C0) { base = p.in0 + cc; Cs = p.C0; } else { base = p.in1 + (cc - p.C0); Cs = p.C1; }
#pragma unroll
        for (int k = 0; k < MAXW; ++k) {
            const int sp = srcT[k * 512 + tid];
            xr[k] = *reinterpret_cast<const float4*>(base + (size_t)(sp < 0 ? 0 : sp) * Cs);
        }
        if constexpr (MODE != WMODE_NONE) {
            const int o = bs * p.in_bstride + (s << 4) + grp * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                sA[c] = p.inA[o + c]; sB[c] = p.inB[o + c];
                if constexpr (MODE == WMODE_AFFINE_E) sE[c] = p.inE[o + c];
            }
        }
    };
    auto a_store = [&]() {
        if (ABL(2)) return;
#pragma unroll
        for (int k = 0; k < MAXW; ++k) {
            float4 v = xr[k];
            if constexpr (MODE != WMODE_NONE) {
                float* vp = &v.x;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float a = qd ? sA[4 + c] : sA[c], b = qd ? sB[4 + c] : sB[c];
                    float w = silu_f(vp[c] * a + b);
                    if constexpr (MODE == WMODE_AFFINE_E) w += qd ? sE[4 + c] : sE[c];
                    vp[c] = w;
                }
            }
            if (srcT[k * 512 + tid] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding applies to the transformed tensor
            *reinterpret_cast<float4*>(W2 + dstT[k * 512 + tid]) = v;
        }
    };

    // ---- T part: thread = (tile tm, channel pair cpq of this group's plane); reads its 4 x 4 patch, writes 16 positions
    auto t_part = [&](char* Vn) {
        if (ABL(16)) return;
        int uu = u;
        asm volatile("" : "+v"(uu));                   // addresses are recomputed here every stage (a handful of VALU ops), not kept live
        const int tm = uu >> 2, cpq = uu & 3;
        const int t_src = grp * PLANE + ((4 * (tm >> 3)) * 9 + (tm & 7)) * 32 + cpq * 8;
        const int t_sw = (tm >> 2) & 3;
        const int t_hi = tm * 64 + ((grp ^ t_sw) << 4) + cpq * 4;
        const int t_lo = tm * 64 + (((2 + grp) ^ t_sw) << 4) + cpq * 4;
        // One channel at a time (16 patch registers, not 32).  hi / lo of two POSITIONS share a cvt_pk; the first channel's
        // results wait packed (position pair per register) until the second channel's arrive, then byte permutes build the
        // [channel 0 | channel 1] words the MFMA fragments want.
        unsigned keep_h[8], keep_l[8];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            float d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[r][c] = *reinterpret_cast<const float*>(W2 + t_src + ((2 * r + (c & 1)) * 9 + (c >> 1)) * 32 + ch * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // column j of d B: (d B)[r][j] = d[r][c0] +- d[r][c1] with (c0, c1, sign) = (0,2,-), (1,2,+), (2,1,-), (1,3,-)
                constexpr int C0[4] = {0, 1, 2, 1}, C1[4] = {2, 2, 1, 3};
                float t[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = j == 1 ? d[r][C0[j]] + d[r][C1[j]] : d[r][C0[j]] - d[r][C1[j]];
                const float vv[4] = {t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]};     // B^T (d B): positions (0..3, j)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {       // position pair (2 ip, j), (2 ip + 1, j)
                    unsigned lo2;
                    const unsigned hi2 = pack_hi_lo(vv[2 * ip], vv[2 * ip + 1], lo2);
                    const int slot = j * 2 + ip;
                    if (ch == 0) { keep_h[slot] = hi2; keep_l[slot] = lo2; }
                    else {
                        char* row0 = Vn + ((2 * ip) * 4 + j) * 4096, *row1 = Vn + ((2 * ip + 1) * 4 + j) * 4096;
                        *reinterpret_cast<unsigned*>(row0 + t_hi) = __builtin_amdgcn_perm(hi2, keep_h[slot], 0x05040100u);   // low halves: ch0 | ch1
                        *reinterpret_cast<unsigned*>(row0 + t_lo) = __builtin_amdgcn_perm(lo2, keep_l[slot], 0x05040100u);
                        *reinterpret_cast<unsigned*>(row1 + t_hi) = __builtin_amdgcn_perm(hi2, keep_h[slot], 0x07060302u);   // high halves
                        *reinterpret_cast<unsigned*>(row1 + t_lo) = __builtin_amdgcn_perm(lo2, keep_l[slot], 0x07060302u);
                    }
                }
            }
        }
    };

    // ---- M part: A fragments from V (rows = tiles), B fragments = this wave's weight stream
    const int m_sw = (l31 >> 2) & 3;
    const int m_hi = l31 * 64 + ((half ^ m_sw) << 4);
    const int m_lo = l31 * 64 + (((2 + half) ^ m_sw) << 4);
    const int NB = p.CoutPad >> 5;
    const unsigned short* wbase = p.wino + ((size_t)(wi * 4) * NB + (n0 >> 5) + nb) * 1024 + lane * 8;
    const size_t w_pos = (size_t)NB * 1024, w_stage = 16 * w_pos;       // shorts between positions / stages
    uint4 bh0 = make_uint4(0, 0, 0, 0), bh1 = bh0, bh2 = bh0, bh3 = bh0, bl0 = bh0, bl1 = bh0, bl2 = bh0, bl3 = bh0;
#define HD_WLOAD(j, s)                                                                                  \
    if (!ABL(4)) {                                                                                      \
        const unsigned short* g = wbase + (size_t)(s) * w_stage + (j) * w_pos;                          \
        bh##j = *reinterpret_cast<const uint4*>(g);                                                     \
        bl##j = *reinterpret_cast<const uint4*>(g + 512);                                               \
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][mb][r] = 0.f;
#define HD_MPOS(j, Vc)                                                                                  \
    if (!ABL(8)) {                                                                                      \
        const char* a = (Vc) + (wi * 4 + (j)) * 4096;                                                   \
        const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(a + m_hi), al0 = *reinterpret_cast<const bf16x8*>(a + m_lo);             \
        const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(a + 2048 + m_hi), al1 = *reinterpret_cast<const bf16x8*>(a + 2048 + m_lo); \
        const bf16x8 wh = __builtin_bit_cast(bf16x8, bh##j), wl = __builtin_bit_cast(bf16x8, bl##j);    \
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, wh, acc[j][0], 0, 0, 0);               \
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, wh, acc[j][1], 0, 0, 0);               \
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, wl, acc[j][0], 0, 0, 0);               \
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, wl, acc[j][1], 0, 0, 0);               \
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, wh, acc[j][0], 0, 0, 0);               \
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, wh, acc[j][1], 0, 0, 0);               \
    }
// positions 0, 1: the next stage's fragments are requested as the registers fall free; positions 2, 3: at the end of the T part that
// precedes the next M part (their 16 registers are then free while the T part is at its register peak)
#define HD_MSTAGE(Vc, sn)                                                                               \
    HD_MPOS(0, Vc) HD_WLOAD(0, sn) HD_MPOS(1, Vc) HD_WLOAD(1, sn) HD_MPOS(2, Vc) HD_MPOS(3, Vc)

    // ---- prologue: window 0 -> V[0]; group 0's plane of window 1 staged; slices 2 (group 0) / 1 (group 1) and the weights
    // of stage 0 in flight.  Group g stores its plane of window s + lead in stage s: lead = 2 - g.
    const int lead = 2 - grp;
    a_request(0);
    HD_WLOAD(0, 0) HD_WLOAD(1, 0) HD_WLOAD(2, 0) HD_WLOAD(3, 0)
    a_store();
    __syncthreads();
    t_part(Vs);
    __syncthreads();
    if (grp == 0) { a_request(1); a_store(); }
    a_request(lead);

    // One code path per group (a wave-uniform branch taken once; both paths run the same number of barriers): group 0 is in the
    // T role in the first half of a stage and in the M role in the second, group 1 the other way round.
    float* const S = reinterpret_cast<float*>(smem);
    auto run = [&](auto first_is_t) {
        constexpr bool T_FIRST = decltype(first_is_t)::value;
#ifdef HD_DIAG
        unsigned long long cT = 0, cM = 0, cB1 = 0, cB2 = 0, t0, t1;
        const unsigned long long tstart = STAMP();
#endif
        for (int s = 0; s < ns; ++s) {
            const char* Vc = Vs + (s & 1) * VBUF;
            char* Vn = Vs + ((s + 1) & 1) * VBUF;
            const int sn = s + 1 < ns ? s + 1 : ns - 1;
#ifdef HD_DIAG
            t0 = STAMP();
#endif
            __syncthreads();     // B1: V[s] complete; plane 0 holds window s+1; plane 1's readers are done
#ifdef HD_DIAG
            t1 = STAMP(); cB1 += t1 - t0;
#endif
            if constexpr (T_FIRST) {
                t_part(Vn);
                HD_WLOAD(2, s) HD_WLOAD(3, s)
            } else {
                a_store();                             // plane 1 of window s+1
                a_request(s + 2);
                HD_MSTAGE(Vc, sn)
            }
#ifdef HD_DIAG
            t0 = STAMP(); if (T_FIRST) cT += t0 - t1; else cM += t0 - t1;
#endif
            __syncthreads();     // Bmid: plane 0's readers are done; plane 1 holds window s+1
#ifdef HD_DIAG
            t1 = STAMP(); cB2 += t1 - t0;
#endif
            if constexpr (T_FIRST) {
                a_store();                             // plane 0 of window s+2
                a_request(s + 3);
                HD_MSTAGE(Vc, sn)
            } else {
                t_part(Vn);
                HD_WLOAD(2, sn) HD_WLOAD(3, sn)
            }
#ifdef HD_DIAG
            t0 = STAMP(); if (T_FIRST) cM += t0 - t1; else cT += t0 - t1;
#endif
        }
        __syncthreads();                               // all MFMA operand reads done: V becomes the exchange buffer
#ifdef HD_DIAG
        const unsigned long long tloop = STAMP();
#endif
        // ---- epilogue 1: output transform along j in registers, T[b] -> S[i][b][tile][channel]
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const f32x16 t0v = acc[0][mb] + acc[1][mb] + acc[2][mb];
            const f32x16 t1v = acc[1][mb] - acc[2][mb] - acc[3][mb];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tile = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int col = nb * 32 + l31;
                S[((wi * 2 + 0) * 64 + tile) * 64 + col] = t0v[r];
                S[((wi * 2 + 1) * 64 + tile) * 64 + col] = t1v[r];
            }
        }
#ifdef HD_DIAG
        if (blockIdx.x == 5 && (tid & 255) == 0) {
            unsigned long long* g = g_wino_stamps[T_FIRST ? 0 : 1];
            g[0] = cT; g[1] = cM; g[2] = cB1; g[3] = cB2; g[4] = tloop - tstart; g[5] = STAMP() - tloop;
        }
#endif
    };
    if (grp == 0) run(std::true_type{}); else run(std::false_type{});
#undef HD_MSTAGE
#undef HD_MPOS
#undef HD_WLOAD
    __syncthreads();

    // ---- epilogue 2: thread = (tile t0 / t0 + 32, channel quad cq); transform along i, then the convolution's epilogue
    const int cq = tid & 15, t0i = tid >> 4;
    const int n = n0 + cq * 4;
    const bool nok = n < p.Cout;                       // Cout is a multiple of 4 here (checked by the launcher)
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && nok) bias = *reinterpret_cast<const float4*>(p.bias + n);
    float4 fsc = make_float4(0.f, 0.f, 0.f, 0.f), fsh = fsc;
    if ((p.ep & (EP_FILM_SILU | EP_ADD_SILU)) && nok) {
        const int fo = bs * p.ep_bstride + n;
        fsh = *reinterpret_cast<const float4*>(p.epShift + fo);
        if (p.ep & EP_FILM_SILU) fsc = *reinterpret_cast<const float4*>(p.epScale + fo);
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int tile = t0i + 32 * it;
        const int ty = tile >> 3, tx = tile & 7;
        float4 q[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b) q[i][b] = *reinterpret_cast<const float4*>(S + ((i * 2 + b) * 64 + tile) * 64 + cq * 4);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v[4];
                if (a == 0) {
                    v[0] = q[0][b].x + q[1][b].x + q[2][b].x; v[1] = q[0][b].y + q[1][b].y + q[2][b].y;
                    v[2] = q[0][b].z + q[1][b].z + q[2][b].z; v[3] = q[0][b].w + q[1][b].w + q[2][b].w;
                } else {
                    v[0] = q[1][b].x - q[2][b].x - q[3][b].x; v[1] = q[1][b].y - q[2][b].y - q[3][b].y;
                    v[2] = q[1][b].z - q[2][b].z - q[3][b].z; v[3] = q[1][b].w - q[2][b].w - q[3][b].w;
                }
                v[0] += bias.x; v[1] += bias.y; v[2] += bias.z; v[3] += bias.w;
                const int y = Y0 + 2 * ty + a, x = X0 + 2 * tx + b;
                if (!nok || y >= p.H || x >= p.W) continue;
#pragma unroll
                for (int c = 0; c < 4; ++c) { s1[c] += v[c]; s2[c] += v[c] * v[c]; }       // GroupNorm sums: the pre-activation output
                const size_t o = ((size_t)(bs * p.H + y) * p.W + x) * p.Cout + n;
                if (p.ep & (EP_FILM_SILU | EP_ADD_SILU)) {
                    if (p.ep & EP_FILM_SILU) {
                        v[0] = v[0] * (fsc.x + 1.f) + fsh.x; v[1] = v[1] * (fsc.y + 1.f) + fsh.y;
                        v[2] = v[2] * (fsc.z + 1.f) + fsh.z; v[3] = v[3] * (fsc.w + 1.f) + fsh.w;
                    } else { v[0] += fsh.x; v[1] += fsh.y; v[2] += fsh.z; v[3] += fsh.w; }
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = silu_f(v[c]);
                }
                if (p.ep & EP_RES) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.res + o);
                    v[0] = p.alpha * v[0] + rr.x; v[1] = p.alpha * v[1] + rr.y; v[2] = p.alpha * v[2] + rr.z; v[3] = p.alpha * v[3] + rr.w;
                }
                const f32x4 o4 = {v[0], v[1], v[2], v[3]};
                if (!ABL(1)) *reinterpret_cast<f32x4*>(p.out + o) = o4;
            }
    }
    if (p.gn_part) {
        // per-channel (sum, sum of squares) of this block's 256 pixels, reduced over the 32 threads of a channel quad in a
        // fixed order: one slot per 16 x 16 block
        float* red = reinterpret_cast<float*>(W2);     // [32][64][2]; the window is dead
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[(t0i * 64 + cq * 4 + c) * 2] = s1[c];
            red[(t0i * 64 + cq * 4 + c) * 2 + 1] = s2[c];
        }
        __syncthreads();
        if (tid < 64 && n0 + tid < p.Cout) {
            float a = 0.f, b = 0.f;
            for (int g = 0; g < 32; ++g) { a += red[(g * 64 + tid) * 2]; b += red[(g * 64 + tid) * 2 + 1]; }
            const int slot = by * p.tiles_x + bx;
            float* dd = p.gn_part + (((size_t)bs * p.gn_slots + slot) * p.Cout + n0 + tid) * 2;
            dd[0] = a; dd[1] = b;
        }
    }
}

// packed fp32 [9 taps][Cin][CoutPad] -> U = G g G^T (fp64), split into bf16 hi / lo, in MFMA B-fragment order:
// [stage = ci / 16][position i*4+j][32-column block][hi | lo][lane = 32 * ((ci % 16) / 8) + co % 32][8 = ci % 8]
__global__ __launch_bounds__(256) void pack_winograd_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int Cin, int CoutPad) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)Cin * CoutPad) return;
    const int co = (int)(idx % CoutPad), ci = (int)(idx / CoutPad);
    double g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[((size_t)(ky * 3 + kx) * Cin + ci) * CoutPad + co];
    const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    const int NB = CoutPad >> 5, s = ci >> 4, kk = ci & 15, ln = (kk >> 3) * 32 + (co & 31), e = kk & 7, nbg = co >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double u = 0.0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) u += G[i][ky] * G[j][kx] * g[ky][kx];
            const float uf = (float)u;
            const __bf16 hi = (__bf16)uf;
            const __bf16 lo = (__bf16)(uf - (float)hi);
            const size_t base = ((((size_t)s * 16 + (i * 4 + j)) * NB + nbg) * 2) * 512 + ln * 8 + e;
            dst[base] = __builtin_bit_cast(unsigned short, hi);
            dst[base + 512] = __builtin_bit_cast(unsigned short, lo);
        }
}

template <typename K>
int launch_w(K kernel, const char* name, ConvLaunch& L, hipStream_t st) {
    static std::set<const void*> raised;
    if (!raised.count(reinterpret_cast<const void*>(kernel))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hd_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); return -3;
        }
        raised.insert(reinterpret_cast<const void*>(kernel));
    }
    const ConvKArgs& k = L.k;
    dim3 grid((unsigned)(k.B * k.tiles_y * k.tiles_x * k.ntiles_n));
    conv_prof_begin(L, name, st);
    hipLaunchKernelGGL(kernel, grid, dim3(512), L.lds, st, L.k);
    conv_prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("winograd conv launch: ") + hipGetErrorString(e)); return -3; }
    return 0;
}

}  // namespace

size_t conv_winograd_weight_bytes(int Cin, int CoutPad) { return (size_t)Cin * CoutPad * 16 * 2 * sizeof(unsigned short); }

int launch_pack_winograd(const float* packed, unsigned short* dst, int Cin, int CoutPad, hipStream_t st) {
    const size_t total = (size_t)Cin * CoutPad;
    hipLaunchKernelGGL(pack_winograd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, packed, dst, Cin, CoutPad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("pack_winograd: ") + hipGetErrorString(e)); return -3; }
    return 0;
}

int launch_conv_winograd(ConvLaunch& L, hipStream_t st) {
    L.lds = (size_t)LDS_BYTES;
    const int mode = L.k.in_mode == IN_AFFINE_SILU ? (L.k.inE ? WMODE_AFFINE_E : WMODE_AFFINE) : WMODE_NONE;
    const bool up = L.k.upsample != 0;
#define HD_W(M, U) return launch_w(conv_winograd_bf16x3_kernel<M, U>, "conv_winograd_bf16x3_kernel<" #M ", " #U ">", L, st)
    if (!up) {
        if (mode == WMODE_NONE) HD_W(0, 0);
        if (mode == WMODE_AFFINE) HD_W(1, 0);
        HD_W(2, 0);
    }
    if (mode == WMODE_NONE) HD_W(0, 1);
    if (mode == WMODE_AFFINE) HD_W(1, 1);
    HD_W(2, 1);
#undef HD_W
}

#ifdef HD_DIAG
extern "C" int hd_debug_wino_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wino_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -3;
}
#endif
