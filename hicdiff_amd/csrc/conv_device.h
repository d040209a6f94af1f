// Device-side pieces shared by the two implicit-GEMM convolution kernels (conv_f32.hip: exact fp32
// MFMA; conv_bf16x3.hip: split-bf16 x3 MFMA): launch arguments, tile decoding, the staged-pixel
// tables, the loader transform and the epilogue.
#pragma once
#include "hd_common.h"

#ifndef HD_CONV_GLDS
#define HD_CONV_GLDS 1    // bf16x3 3x3 kernels: weights by LDS-DMA (global_load_lds_dwordx4) into a ring of three slabs; 0: through prefetch registers +
                          // ds_write_b128 into two (the round-2 path; A/B builds: make TAG=_g0 EXTRA=-DHD_CONV_GLDS=0).  The host's LDS plan follows it.
#endif

// ... per kernel variant (round 4).  profiles/r03_d measured the register path ahead on exactly two variants of the 256 x 64 tile at 64 x 64
// maps -- plain loader with 16-channel slices (214 vs 222-224 us on 64 -> 64) and the GroupNorm-apply loader with 32-channel slices (262 vs
// 278-299 us) -- and level or behind everywhere else.  Those two take the register path; HD_CONV_GLDS=2 forces LDS-DMA on every variant (A/B).
// mode: the kernel's MODE template argument (IN_* ; IN_AFFINE_SILU_E = 3 keeps LDS-DMA: not measured apart).
constexpr bool conv_glds(int wm, int wn, int ck, int mode, int ntaps) {
    if (ntaps != 9 || HD_CONV_GLDS == 0) return false;
    if (HD_CONV_GLDS == 1 && wm == 4 && wn == 1 && ((ck == 16 && mode == 0 /* IN_NONE */) || (ck == 32 && mode == 1 /* IN_AFFINE_SILU */))) return false;
    return true;
}

#ifndef HD_GLDS_D
#define HD_GLDS_D 1       // LDS-DMA path: slabs requested ahead of the one being read (ring = HD_GLDS_D + 1 slabs); 1 and 2 measure the same (profiles/r03_*), 1 needs a third less LDS
#endif
#ifndef HD_MFMA16
#define HD_MFMA16 1       // bf16x3 kernels with 32-channel slices: v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (same flops per cycle; the chip holds a higher clock on it)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvKArgs {
    const float* in0; const float* in1;
    int C0, C1, Cin;
    int B, H, W, IH, IW, stride, pad, upsample, KH, KW;
    const float* w; const unsigned short* wsplit; const unsigned short* wino; const float* bias;
    int Cout, CoutPad;
    int TB, TH, TW, LH, LW, npx, tiles_x, tiles_y, ntiles_n;
    int xs_stride;               // bf16x3 kernel: bytes between its two activation windows in LDS (0: single window)
    int pt_n4;                   // bf16x3 kernel: 16-byte entries per vector of the loader-parameter table (TB * CK / 4; LayerNorm: CK / 4)
    int in_mode; const float* inA; const float* inB; const float* inE; int in_bstride;
    const float* ln_stats; const float* ln_g;
    int ep; const float* epScale; const float* epShift; int ep_bstride;
    float alpha; const float* res; const float* resA; const float* resB; int res_bstride;
    const float* ep_ln_g; unsigned long long w_bstride; float* ln_stats_out;
    float* out; float* pre_out; float* gn_part; int gn_slots;
    GnFinArgs fin; int gn_direct;   // gn_direct: the workgroup holds every pixel of its sample(s): write the GroupNorm affine (fin) instead of leaving it to gn_finalize
    int ksplit, kchunks;         // bf16x3 3x3 kernel: split-K over grid.y (1: off); K slices per split
    unsigned long long split_stride;   // floats between the splits' partial outputs (out then points at the workspace)
    int m16;      // bf16x3 kernel family: 1 = the launch runs on 16 x 16 MFMA tiles (row -> pixel maps are laid out for 16-row operand blocks)
    int f16w2;    // bf16x3 kernel family, 3x3 kernels: 1 = two fp16 products per multiply, xh (wh + wl); 2 = one, xh wh; wsplit then points at the fp16 image
    int plain;    // bf16x3 kernel family: 1 = one bf16 MFMA per product (training's optional bf16 arithmetic: 32-channel-slice kernels, plain / GroupNorm-apply loaders)
#ifdef HD_STAMPS
    unsigned long long* stamps;   // timing study builds only (make EXTRA=-DHD_STAMPS): [4096][16] cycle stamps, see conv_bf16x3_kernel.h
#endif
    int ablate;   // timing experiments only (HICDIFF_ABLATE): 1 no epilogue stores, 2 no X staging, 4 no W staging, 8 no MFMA
};

// x * sigmoid(x) with the hardware exp2 / reciprocal (relative error ~1e-7)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

struct TileCtx {
    int tid, lane, wm, wn, half, l31;
    int tile_x, tile_y, b0, y0, x0, n0;
    int thw, mvalid;
};

// 4 waves arranged WM x WN; wave (wm, wn) owns rows [wm*32*TM, +32*TM) and columns [wn*32*TN, +32*TN).
template <int WN, int BN>
__device__ __forceinline__ TileCtx tile_decode(const ConvKArgs& p) {
    TileCtx t;
    t.tid = threadIdx.x; t.lane = t.tid & 63;
    const int wave = t.tid >> 6;
    t.wm = wave / WN; t.wn = wave % WN; t.half = t.lane >> 5; t.l31 = t.lane & 31;
    // XCD-aware order: hardware deals workgroups round-robin over the 8 XCDs (private L2 each); remap so
    // each XCD gets a CONTIGUOUS run of logical tiles -- the N-tiles of one M-tile and neighbouring
    // M-tiles (shared halo rows, same weights) then share an L2.  Bijective: the tail keeps its index.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, per = nwg >> 3;
        if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
    }
    const int nt = bid % p.ntiles_n;
    int mt = bid / p.ntiles_n;
    t.tile_x = mt % p.tiles_x; mt /= p.tiles_x;
    t.tile_y = mt % p.tiles_y;
    const int tile_b = mt / p.tiles_y;
    t.b0 = tile_b * p.TB; t.y0 = t.tile_y * p.TH; t.x0 = t.tile_x * p.TW; t.n0 = nt * BN;
    t.thw = p.TH * p.TW; t.mvalid = p.TB * t.thw;
    return t;
}

// GEMM row m -> output pixel (tb, ty, tx) of the tile.  Any bijection works as long as the operand reads and
// the output table agree; when the window is 16 pixels wide, pixel row ty is rotated by ty*(LW-16) so that
// the staged-pixel index of MFMA row r is congruent to r modulo 16: the 16 lanes of every ds_read_b128
// group then hit 16 distinct 16-byte bank slots although a 32-row block spans two pixel rows.
__device__ __forceinline__ void decode_row(const ConvKArgs& p, int thw, int m, int& tb, int& ty, int& tx) {
    tb = m / thw;
    const int r = m - tb * thw;
    ty = r / p.TW;
    tx = r - ty * p.TW;
    if (p.TW == 16 && p.stride == 1) tx = (tx - ty * (p.LW - 16)) & 15;
    // 8 x 8 tiles of a 3x3 filter (window 10 wide): a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27} and
    // {4-11, 16-19, 28-31} (MI355X_MICROARCH.md, LDS), and a group is conflict-free when its 16 staged indices 10*ty + tx are
    // distinct modulo 16.  Pixel rows t and t + 4 together cover every residue once (10*4 = 8 mod 16), so each lane group takes
    // one such pair: block 0 of an image = rows {0, 4} | {2, 6}, block 1 = rows {1, 5} | {3, 7}.  (Round 1 paired rows g, g + 4
    // over 16 CONSECUTIVE lanes, which the real lane groups cut across: the SQ counters showed a quarter of these layers' LDS
    // cycles as bank conflicts.)
    if (p.TW == 8 && p.TH == 8 && p.LW == 10 && p.stride == 1 && p.m16) {
        // 16 x 16 MFMA tiles: an operand block is 16 GEMM rows = pixel rows j and j + 4 (10 * 4 = 8 mod 16: the block's 16 staged indices are
        // consecutive modulo 16, which is what the 160-byte pitch of that path needs -- conv_bf16x3_kernel.h)
        const int j = r >> 4, l = r & 15;
        ty = j + 4 * (l >> 3);
        tx = l & 7;
    } else if (p.TW == 8 && p.TH == 8 && p.LW == 10 && p.stride == 1) {
        const int blk = r >> 5, l = r & 31;
        const bool inA = l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28);
        const int k = inA ? (l < 4 ? l : l < 16 ? l - 8 : l - 12) : (l < 12 ? l - 4 : l < 20 ? l - 8 : l - 16);   // position inside its group
        ty = blk + (inA ? 0 : 2) + 4 * (k >> 3);
        tx = k & 7;
    }
}

// pxsrc[i]: linear index of the stored input pixel feeding staged pixel i (or -1: zero padding /
// outside the batch); rowpix[m]: linear output pixel of GEMM row m (or -1).
template <int BM, int NT>
__device__ __forceinline__ void init_tables(const ConvKArgs& p, const TileCtx& t, int* pxsrc, int* pxb, int* rowpix, int* rowb) {
    const int LH = p.LH, LW = p.LW;
    for (int i = t.tid; i < p.npx; i += NT) {
        int tb = i / (LH * LW);
        int r = i - tb * LH * LW;
        int ly = r / LW, lx = r - ly * LW;
        int b = t.b0 + tb;
        int iy = t.y0 * p.stride + ly - p.pad, ix = t.x0 * p.stride + lx - p.pad;
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
    for (int m = t.tid; m < BM; m += NT) {
        int tb, ty, tx;
        decode_row(p, t.thw, m, tb, ty, tx);
        int b = t.b0 + tb, y = t.y0 + ty, x = t.x0 + tx;
        bool v = (m < t.mvalid) && b < p.B && y < p.H && x < p.W;
        rowpix[m] = v ? (b * p.H + y) * p.W + x : -1;
        rowb[m] = b < p.B ? b : p.B - 1;
    }
}

// staged-pixel offset of GEMM row m inside the LDS window
__device__ __forceinline__ int row_px_offset_m(const ConvKArgs& p, const TileCtx& t, int m) {
    if (m >= t.mvalid) m = 0;
    int tb, ty, tx;
    decode_row(p, t.thw, m, tb, ty, tx);
    return (tb * p.LH + ty * p.stride) * p.LW + tx * p.stride;
}
// ... of GEMM row (wm, tm, lane) of the 32 x 32 tiling
template <int TM>
__device__ __forceinline__ int row_px_offset(const ConvKArgs& p, const TileCtx& t, int tm) {
    int m = t.wm * 32 * TM + tm * 32 + t.l31;
    if (m >= t.mvalid) m = 0;
    int tb, ty, tx;
    decode_row(p, t.thw, m, tb, ty, tx);
    return (tb * p.LH + ty * p.stride) * p.LW + tx * p.stride;
}

// loader transform on 4 consecutive channels (starting at cc) of staged pixel s of sample b
__device__ __forceinline__ float4 transform4(const ConvKArgs& p, float4 v, int cc, int s, int b) {
    if (p.in_mode == IN_AFFINE_SILU) {
        const int o = b * p.in_bstride + cc;
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
        const float4 g = *reinterpret_cast<const float4*>(p.ln_g + cc);
        v.x = (v.x - mu) * rs * g.x; v.y = (v.y - mu) * rs * g.y;
        v.z = (v.z - mu) * rs * g.z; v.w = (v.w - mu) * rs * g.w;
    }
    return v;
}

// Epilogue: the accumulator tile goes through LDS once so that every global access of the epilogue is a
// 16-byte row-contiguous access (a 32x32 MFMA C tile holds one column per lane: stored directly that is
// 64 four-byte stores per lane plus per-element address arithmetic, which measured ~35 % of the kernel).
//   stage: LDS, BM x (BN + 4) floats, overlays the operand buffers (all MFMA reads are done).
//   thread -> 4 fixed output channels (cq) and rows rg, rg + RPP, ...; GroupNorm per-channel partial sums
//   are accumulated along those rows and reduced over the row groups in a fixed order: one slot per tile.
// Acc: f32x16[TM][TN] (32 x 32 MFMA tiles) or f32x4[2 TM][2 TN] (16 x 16 tiles: col = lane & 15, row = 4 * (lane >> 4) + reg).
// GroupNorm finalize by the producer (GnFinArgs): every lane holds one channel's (sum, sum of squares) over ALL pixels of sample b; lanes of
// a wave hold consecutive channels starting at a multiple of 64 and a group is cg = C / groups channels, a power of two <= 64, so a group
// never straddles a wave.  fp64 butterfly over the group's lanes (a fixed order), then what gn_finalize_kernel (small_kernels.hip) writes:
// A = rstd gamma [(scale + 1)], Bv = (beta - mean rstd gamma) [(scale + 1) + shift], E = SR3's additive term.  Call with whole waves.
__device__ __forceinline__ void gn_direct_finalize(const GnFinArgs& f, int b, int c, int C, int HW, float sum, float sumsq) {
    const int cg = C / f.groups;
    double s1 = sum, s2 = sumsq;
    for (int o = 1; o < cg; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const double n = (double)HW * cg;
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    const int i = b * C + c;
    float a = rstd * f.gamma[c];
    float bb = f.beta[c] - (float)mean * a;
    if (f.film_mode == 1) {
        const float sc = f.film[(size_t)b * f.film_bs + f.film_off + c] + 1.f;
        const float sf = f.film[(size_t)b * f.film_bs + f.film_off + C + c];
        a *= sc; bb = bb * sc + sf;
    } else if (f.film_mode == 2) {
        f.E[i] = f.film[(size_t)b * f.film_bs + f.film_off + c];
    }
    f.A[i] = a; f.Bv[i] = bb;
}

// FBWD: the instantiation carries the training-only EP_FILM_SILU_BWD mode (one kernel of its own, conv_igemm_bf16x3_fbwd_kernel: compiled into the
// sampler's kernels it cost hicedrn64 1.8 % and tipped the register-capped dominant unet64 kernel into scratch -- measured, A/B on one box).
template <int BM, int BN, int TM, int TN, int NT, bool FBWD = false, int EPMASK = -1, typename Acc>
__device__ __forceinline__ void conv_epilogue(const ConvKArgs& p, const TileCtx& t, Acc& acc, const int* rowpix,
                                              const int* rowb, float* stage) {
    constexpr bool M16 = sizeof(acc[0][0]) == 16;
    // EPMASK: the epilogue modes this instantiation can be asked for (the host never sends others: plan_conv); the rest compiles out
    const int ep = p.ep & EPMASK;
    // The tile is staged in TM rounds of BM/TM rows (round tm holds, for every wave-row wm, its rows
    // tm*32..tm*32+31), so the staging area is BM/TM x (BN+4) floats and does not set the LDS footprint.
    constexpr int EP = BN + 4, CQ = BN / 4, RPP = NT / CQ, RB = BM / TM, NPASS = RB / RPP, WMN = BM / (32 * TM);
    const int cq = t.tid % CQ, rg = t.tid / CQ;
    const int n = t.n0 + cq * 4;
    const int nvalid = p.Cout - n;         // >= 4: all four channels of this thread exist
    const bool vec = nvalid >= 4 && (p.Cout & 3) == 0;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        if (vec) {     // one 16-byte load, requested here and first used after the staging round (four guarded scalar loads sat behind a vmcnt(0) right here)
            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
            bias[0] = b4.x; bias[1] = b4.y; bias[2] = b4.z; bias[3] = b4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < nvalid) bias[j] = p.bias[n + j];
        }
    }
    // GroupNorm partial sums of this lane's 4 channels.  Two sets: with two whole 8x8 images per tile (TB == 2, BM == 128)
    // rows 0..63 belong to sample b0 and rows 64..127 to b0+1, i.e. the upper half of every staging round (lr >= 32).
    float s1[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, s2[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const bool two = p.TB == 2;
#ifdef HD_STAMPS
    unsigned long long e_bar = 0, e_stage = 0, e_store = 0, e0, e1;
#endif
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        // Residual rows of the whole round, requested BEFORE the accumulators are staged: fetched pass by pass (one ahead), each of the
        // NPASS passes waited out most of an HBM round trip -- the 1x1 shortcut at 64 x 64 spent two thirds of a workgroup's life there
        // (3.9 TB/s with the residual against 5.0 without).  Rows past the tile read pixel 0 (never used).
        float4 rr_all[NPASS];
        const bool pre_res = vec && nvalid > 0 && (ep & (EP_RES | EP_RES_AFFINE_SILU | EP_LN_RES | (FBWD ? EP_FILM_SILU_BWD : 0))) != 0;
        if (pre_res) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int lr = pass * RPP + rg;
                const int pix = rowpix[(lr >> 5) * 32 * TM + tm * 32 + (lr & 31)];
                rr_all[pass] = *reinterpret_cast<const float4*>(p.res + (size_t)(pix < 0 ? 0 : pix) * p.Cout + n);
            }
        }
#ifdef HD_STAMPS
        e0 = __builtin_readcyclecounter();
#endif
        __syncthreads();                   // operands (tm == 0) / previous round's rows are no longer read
#ifdef HD_STAMPS
        e1 = __builtin_readcyclecounter(); e_bar += e1 - e0;
#endif
        if constexpr (M16) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j) {
                    const int col = t.wn * 32 * TN + j * 16 + (t.lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lr = t.wm * 32 + ii * 16 + 4 * (t.lane >> 4) + r;    // row inside this round
                        stage[lr * EP + col] = acc[2 * tm + ii][j][r];
                    }
                }
        } else {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int col = t.wn * 32 * TN + tn * 32 + t.l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = t.wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * t.half;    // row inside this round
                    stage[lr * EP + col] = acc[tm][tn][r];
                }
            }
        }
#ifdef HD_STAMPS
        e0 = __builtin_readcyclecounter(); e_stage += e0 - e1;
#endif
        __syncthreads();
#ifdef HD_STAMPS
        e1 = __builtin_readcyclecounter(); e_bar += e1 - e0;
#endif
        if (nvalid > 0 && !(p.ablate & 128)) {
            if (ep & EP_LN_RES) {
                // Whole rows live in this workgroup (Cout == BN, checked by the launcher): channel LayerNorm of the
                // row (biased variance, eps 1e-5: src/hicdiff.py:99-108), gain, + residual.  A row is held by CQ
                // consecutive lanes (4 channels each); sums travel by xor shuffles inside that group.
                const float4 g4 = *reinterpret_cast<const float4*>(p.ep_ln_g + n);
                auto rsum = [](float x) {
#pragma unroll
                    for (int m = 1; m < CQ; m <<= 1) x += __shfl_xor(x, m, 64);
                    return x;
                };
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {      // (fully unrolled: rr_all must stay in registers)
                    const int lr = pass * RPP + rg;
                    const int pix = rowpix[(lr >> 5) * 32 * TM + tm * 32 + (lr & 31)];
                    const float4 rr = rr_all[pass];
                    const float4 a4 = *reinterpret_cast<const float4*>(stage + lr * EP + cq * 4);
                    float v[4] = {a4.x + bias[0], a4.y + bias[1], a4.z + bias[2], a4.w + bias[3]};
                    const float mean = rsum(v[0] + v[1] + v[2] + v[3]) * (1.f / BN);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] -= mean;
                    const float var = rsum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]) * (1.f / BN);
                    const float rs = rsqrtf(var + 1e-5f);
                    const f32x4 o4 = {v[0] * rs * g4.x + rr.x, v[1] * rs * g4.y + rr.y, v[2] * rs * g4.z + rr.z, v[3] * rs * g4.w + rr.w};
                    if (pix >= 0) *reinterpret_cast<f32x4*>(p.out + (size_t)pix * p.Cout + n) = o4;
                }
            } else if (ep == 0 && vec) {
                // Fast path (every GroupNorm'd conv): no global load in the loop.  vmcnt retires in issue order, so a
                // loop that mixes loads with stores makes every load wait for the previous pass's STORE round trip
                // (measured: half of the kernel on the 64-channel full-resolution layers); here stores just stream.
                // All passes' LDS reads (row -> pixel table, staged row) are issued first: read one pass at a time, behind
                // the previous pass's guarded store, every pass paid two exposed LDS round trips (HD_STAMPS: 640 cycles per pass, 10 k of a
                // 14 k-cycle epilogue on the 64-channel layers; skipping the stores altogether changed nothing).
                // Instruction count matters here more than anywhere else in the kernel: this wave shares its SIMD with a wave of the CU's
                // other workgroup that is in its MFMA loop, and each VALU instruction waits for an issue slot between that wave's MFMAs.
                int pixs[NPASS];
                float4 rows[NPASS];
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    const int lr = pass * RPP + rg;
                    pixs[pass] = rowpix[(lr >> 5) * 32 * TM + tm * 32 + (lr & 31)];
                    rows[pass] = *reinterpret_cast<const float4*>(stage + lr * EP + cq * 4);
                }
                // One sample per tile (all but the 8x8 maps): one set of sums, no upper / lower-half selects.  The squares are accumulated with an
                // EXPLICIT fma in both branches: under -ffp-contract=fast hipcc fuses `s += x * x` in some instances of this loop and not
                // in others (tools/epilogue_sum_repro.hip: 4 % of the sums differ in the last bit between two equivalent forms, and between the
                // two halves of a two-image tile), and a tile's result must not depend on where the sample sits or on the batch.
                float* const obase = p.out + n;
                const bool keep = !(p.ablate & 1);
                if (!two) {
#pragma unroll
                    for (int pass = 0; pass < NPASS; ++pass) {
                        const int pix = pixs[pass];
                        const float4 a4 = rows[pass];
                        const bool live = pix >= 0;
                        const f32x4 o4 = {a4.x + bias[0], a4.y + bias[1], a4.z + bias[2], a4.w + bias[3]};
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float x = live ? o4[j] : 0.f; s1[0][j] += x; s2[0][j] = __builtin_fmaf(x, x, s2[0][j]); }
                        if (live && keep) *reinterpret_cast<f32x4*>(obase + (unsigned)pix * (unsigned)p.Cout) = o4;
                    }
                } else {
#pragma unroll
                    for (int pass = 0; pass < NPASS; ++pass) {
                        const int lr = pass * RPP + rg;
                        const int pix = pixs[pass];
                        const bool up = lr >= 32;
                        const float4 a4 = rows[pass];
                        const f32x4 o4 = {a4.x + bias[0], a4.y + bias[1], a4.z + bias[2], a4.w + bias[3]};
                        const bool live = pix >= 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {   // selects, not indexing: a dynamically indexed register array goes to scratch
                            const float x = live ? o4[j] : 0.f, lo = up ? 0.f : x, hi = up ? x : 0.f;
                            s1[0][j] += lo; s2[0][j] = __builtin_fmaf(lo, lo, s2[0][j]); s1[1][j] += hi; s2[1][j] = __builtin_fmaf(hi, hi, s2[1][j]);
                        }
                        if (live && keep) *reinterpret_cast<f32x4*>(obase + (unsigned)pix * (unsigned)p.Cout) = o4;
                    }
                }
            } else if (vec) {
                // General path: the residual rows were requested at the top of the round; per-row FiLM operands (several samples per tile only) of
                // pass i+1 are requested BEFORE pass i is stored, so the wait for them never includes the store that was issued after them.
                float4 n_sc = make_float4(0.f, 0.f, 0.f, 0.f), n_sh = n_sc, n_ra = n_sc, n_rb = n_sc;
                // One sample per tile (every full-resolution layer): the per-(sample, channel) operands are the same for all rows --
                // load them once instead of once per pass (vmcnt retires in order: every load in the pass loop is a wait on the stores
                // and loads issued before it).
                const bool one_b = p.TB == 1;
                const bool film_bwd = FBWD && (ep & EP_FILM_SILU_BWD) != 0;
                const bool has_scale = (ep & EP_FILM_SILU) || (film_bwd && p.epScale);
                constexpr int FILM_EPS = EP_FILM_SILU | EP_ADD_SILU | (FBWD ? EP_FILM_SILU_BWD : 0);
                if (one_b) {
                    if (ep & FILM_EPS) {
                        const int fo = t.b0 * p.ep_bstride + n;
                        n_sh = *reinterpret_cast<const float4*>(p.epShift + fo);
                        if (has_scale) n_sc = *reinterpret_cast<const float4*>(p.epScale + fo);
                    }
                    if (ep & EP_RES_AFFINE_SILU) {
                        const int fo = t.b0 * p.res_bstride + n;
                        n_ra = *reinterpret_cast<const float4*>(p.resA + fo);
                        n_rb = *reinterpret_cast<const float4*>(p.resB + fo);
                    }
                }
                auto fetch = [&](int pass) {
                    const int lr = pass * RPP + rg;
                    const int m = (lr >> 5) * 32 * TM + tm * 32 + (lr & 31);
                    if (!one_b && (ep & FILM_EPS)) {
                        const int fo = rowb[m] * p.ep_bstride + n;
                        n_sh = *reinterpret_cast<const float4*>(p.epShift + fo);
                        if (has_scale) n_sc = *reinterpret_cast<const float4*>(p.epScale + fo);
                    }
                    if (!one_b && (ep & EP_RES_AFFINE_SILU)) {
                        const int fo = rowb[m] * p.res_bstride + n;
                        n_ra = *reinterpret_cast<const float4*>(p.resA + fo);
                        n_rb = *reinterpret_cast<const float4*>(p.resB + fo);
                    }
                };
                fetch(0);
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {      // (fully unrolled: rr_all must stay in registers)
                    const int lr = pass * RPP + rg;
                    const int m = (lr >> 5) * 32 * TM + tm * 32 + (lr & 31);
                    const int pix = rowpix[m];
                    const bool up = two && lr >= 32;
                    const float4 rr = rr_all[pass], sc = n_sc, sh = n_sh, ra = n_ra, rb = n_rb;
                    if (pass + 1 < NPASS) fetch(pass + 1);
                    const float4 a4 = *reinterpret_cast<const float4*>(stage + lr * EP + cq * 4);
                    float v[4] = {a4.x + bias[0], a4.y + bias[1], a4.z + bias[2], a4.w + bias[3]};
                    if (film_bwd) {
                        // training: the gradient through silu(u (scale + 1) + shift), and the FiLM row's own gradient sums in the GroupNorm sums' place
                        const float uu[4] = {rr.x, rr.y, rr.z, rr.w}, scv[4] = {sc.x + 1.f, sc.y + 1.f, sc.z + 1.f, sc.w + 1.f}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float vv = uu[j] * scv[j] + shv[j];
                            const float sg = 1.f / (1.f + __expf(-vv));
                            const float dv = p.alpha * v[j] * (sg * (1.f + vv * (1.f - sg)));
                            const float a_ = pix >= 0 ? dv * uu[j] : 0.f, b_ = pix >= 0 ? dv : 0.f;
                            s1[0][j] += up ? 0.f : a_; s2[0][j] += up ? 0.f : b_; s1[1][j] += up ? a_ : 0.f; s2[1][j] += up ? b_ : 0.f;
                            v[j] = dv * scv[j];
                        }
                    } else if (p.gn_part && pix >= 0) {          // (no GroupNorm follows the shortcut / projection convolutions that take this path: skip the sums)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x = v[j], lo = up ? 0.f : x, hi = up ? x : 0.f;
                            s1[0][j] += lo; s2[0][j] = __builtin_fmaf(lo, lo, s2[0][j]); s1[1][j] += hi; s2[1][j] = __builtin_fmaf(hi, hi, s2[1][j]);   // explicit fma: see the fast path
                        }
                    }
                    if (ep & (EP_FILM_SILU | EP_ADD_SILU)) {
                        if (p.pre_out && pix >= 0) *reinterpret_cast<f32x4*>(p.pre_out + (size_t)pix * p.Cout + n) = f32x4{v[0], v[1], v[2], v[3]};
                        if (ep & EP_FILM_SILU) {
                            v[0] = v[0] * (sc.x + 1.f) + sh.x; v[1] = v[1] * (sc.y + 1.f) + sh.y;
                            v[2] = v[2] * (sc.z + 1.f) + sh.z; v[3] = v[3] * (sc.w + 1.f) + sh.w;
                        } else { v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w; }
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
                    }
                    if (ep & EP_RES) { v[0] = p.alpha * v[0] + rr.x; v[1] = p.alpha * v[1] + rr.y; v[2] = p.alpha * v[2] + rr.z; v[3] = p.alpha * v[3] + rr.w; }
                    if (ep & EP_RES_AFFINE_SILU) {
                        v[0] += silu_f(rr.x * ra.x + rb.x); v[1] += silu_f(rr.y * ra.y + rb.y);
                        v[2] += silu_f(rr.z * ra.z + rb.z); v[3] += silu_f(rr.w * ra.w + rb.w);
                    }
                    const f32x4 o4 = {v[0], v[1], v[2], v[3]};
                    if (pix >= 0 && !(p.ablate & 1)) *reinterpret_cast<f32x4*>(p.out + (size_t)pix * p.Cout + n) = o4;
                    if (ep & EP_LN_STATS) {
                        // statistics of the finished row for the PreNorm of the attention block that consumes it (Cout == BN:
                        // the row is the CQ consecutive lanes of this pass); same two-pass form as ln_stats_kernel
                        float sm = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
                        for (int mk = CQ / 2; mk >= 1; mk >>= 1) sm += __shfl_xor(sm, mk, 64);
                        const float mean = sm * (1.f / BN);
                        float q = (v[0] - mean) * (v[0] - mean) + (v[1] - mean) * (v[1] - mean) + (v[2] - mean) * (v[2] - mean) + (v[3] - mean) * (v[3] - mean);
#pragma unroll
                        for (int mk = CQ / 2; mk >= 1; mk >>= 1) q += __shfl_xor(q, mk, 64);
                        if (cq == 0 && pix >= 0) { p.ln_stats_out[2 * (size_t)pix] = mean; p.ln_stats_out[2 * (size_t)pix + 1] = 1.f / sqrtf(q * (1.f / BN) + 1e-5f); }
                    }
                }
            } else {
                // narrow outputs (Cout not a multiple of 4: the 1-channel tail of hicedrn): scalar accesses
                for (int pass = 0; pass < NPASS; ++pass) {
                    const int lr = pass * RPP + rg;
                    const int m = (lr >> 5) * 32 * TM + tm * 32 + (lr & 31);
                    const int pix = rowpix[m];
                    if (pix < 0) continue;
                    const float4 a4 = *reinterpret_cast<const float4*>(stage + lr * EP + cq * 4);
                    const float v[4] = {a4.x + bias[0], a4.y + bias[1], a4.z + bias[2], a4.w + bias[3]};
                    const size_t o = (size_t)pix * p.Cout + n;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j >= nvalid) break;
                        float x = v[j];
                        s1[0][j] += x; s2[0][j] += x * x;
                        if (ep & (EP_FILM_SILU | EP_ADD_SILU)) {
                            const int fo = rowb[m] * p.ep_bstride + n + j;
                            x = (ep & EP_FILM_SILU) ? x * (p.epScale[fo] + 1.f) + p.epShift[fo] : x + p.epShift[fo];
                            x = silu_f(x);
                        }
                        if (ep & EP_RES) x = p.alpha * x + p.res[o + j];
                        if (ep & EP_RES_AFFINE_SILU) {
                            const int fo = rowb[m] * p.res_bstride + n + j;
                            x += silu_f(p.res[o + j] * p.resA[fo] + p.resB[fo]);
                        }
                        p.out[o + j] = x;
                    }
                }
            }
        }
#ifdef HD_STAMPS
        e_store += __builtin_readcyclecounter() - e1;
#endif
    }
    (void)WMN;
    if (p.gn_part) {   // TB == 1: every row of this workgroup belongs to sample b0; TB == 2 (two 8x8 images): a second set
        float* red = stage;
        for (int h = 0; h < (two ? 2 : 1); ++h) {
            __syncthreads();               // stage is free again: reuse it as [RPP][BN][2]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[(rg * BN + cq * 4 + j) * 2] = h ? s1[1][j] : s1[0][j];
                red[(rg * BN + cq * 4 + j) * 2 + 1] = h ? s2[1][j] : s2[0][j];
            }
            __syncthreads();
            if (t.tid < BN && t.n0 + t.tid < p.Cout && t.b0 + h < p.B) {
                float a = 0.f, b = 0.f;
                for (int g = 0; g < RPP; ++g) { a += red[(g * BN + t.tid) * 2]; b += red[(g * BN + t.tid) * 2 + 1]; }
                if (p.gn_direct) {        // (host: Cout % 64 == 0, so the condition above holds or fails for whole waves)
                    gn_direct_finalize(p.fin, t.b0 + h, t.n0 + t.tid, p.Cout, p.H * p.W, a, b);
                } else {
                    const int slot = t.tile_y * p.tiles_x + t.tile_x;
                    float* d = p.gn_part + (((size_t)(t.b0 + h) * p.gn_slots + slot) * p.Cout + t.n0 + t.tid) * 2;
                    d[0] = a; d[1] = b;
                }
            }
        }
    }
#ifdef HD_STAMPS
    if (t.tid == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {
        unsigned long long* g = p.stamps + 16 * blockIdx.x;     // per workgroup (wave 0): cycles inside the epilogue's barriers / LDS staging writes / row passes with their stores
        g[12] = e_bar; g[13] = e_stage; g[14] = e_store;
    }
#endif
}

// host-side launch record shared by the two kernel files
struct ConvLaunch {
    ConvKArgs k;
    size_t lds;
    int ck;          // K slice
    int cfg;         // 0: 128 x 128 tile (waves 2 x 2), 1: 128 x 64 (2 x 2), 2: 256 x 64 (4 x 1), 3: 256 x 128 (8 waves, 4 x 2)
};
int launch_conv_f32(ConvLaunch& L, hipStream_t st);
int launch_conv_bf16x3(ConvLaunch& L, hipStream_t st);
int launch_conv_winograd(ConvLaunch& L, hipStream_t st);   // 16 x 16 pixel blocks of one sample: tiles_y = H / 16, tiles_x = W / 16, ntiles_n = CoutPad / 64
int conv_bf16x3_max_items(int cfg, int ck, bool taps9, bool layernorm);
void conv_prof_begin(const ConvLaunch& L, const char* name, hipStream_t st);   // name: the instantiation as rocprofv3 prints it (string literal)
void conv_prof_end(hipStream_t st);
const char* conv_prof_name(const char* head, int mode, const char* tail);   // interned "head<mode>tail"; nullptr while the profiler is off
