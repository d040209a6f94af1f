// Fused key/value side of LinearAttention (src/hicdiff.py:212-223) for 64-, 128- and 256-channel feature maps:
//
//   LN(x) -> k, v = to_qkv[128:384](LN(x)) -> softmax_n(k) -> context[d][e] = sum_n softmax(k)[d][n] v[e][n] / HW
//
// One workgroup per (sample, 64-token chunk); k and v are never written to HBM (the unfused path wrote a
// 384-channel qkv tensor -- 1.6 GB at B=256, S=64 -- and read two thirds of it back).  Wave w owns head
// w: it multiplies the shared LN'd token tile [64][C] by that head's 64 weight rows (32 k + 32 v) on the
// split-bf16 MFMA path; the accumulators then hold k and v with the token on the REGISTER index and the
// channel on the lane, so the column softmax statistics are in-lane reductions, and the context product,
// which sums over tokens, takes both accumulator tiles directly as MFMA operands (both operands share
// the accumulator layout's k-order, so no lane movement and no LDS round trip).  Each chunk emits
// flash-style partials (column maxima, exp-sums, un-normalised 32x32 context); linattn_combine_kernel
// rescales and normalises them in a fixed order.
#include "hd_common.h"

#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define KV_TOK 64

__device__ __forceinline__ void split8v(const float (&v)[8], uint4& hi, uint4& lo) {
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

// eight consecutive accumulator registers (rows of a 32x32 tile) -> bf16 hi / lo operand fragments
__device__ __forceinline__ void acc_to_frag(const f32x16& a, int s, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = a[8 * s + j];
        const __bf16 hh = (__bf16)v;
        hi[j] = hh;
        lo[j] = (__bf16)(v - (float)hh);
    }
}

// wkv: [C/32 slices][256 rows = head*64 + {k: 0..31, v: 32..63}][32 bf16 hi | 32 bf16 lo], LayerNorm gain folded in.
template <int C>
__global__ __launch_bounds__(256) void linattn_kv_fused_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wkv,
                                                               int HW, int nsplit, float* __restrict__ pmax,
                                                               float* __restrict__ psum, float* __restrict__ pctx) {
    constexpr int CK = 32, PITCH = 4 * CK + 16, ROWB = 4 * CK, NS = C / CK, heads = 4, D = 32;
    constexpr int QPT = C / 4 / 4;                     // float4 per thread: 4 threads share a token
    __shared__ __attribute__((aligned(16))) char Xs[KV_TOK * PITCH];
    __shared__ __attribute__((aligned(16))) char Ws[256 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
    const int n0 = sp * KV_TOK, ntok = min(KV_TOK, HW - n0);

    // ---- LayerNorm of the token rows (4 threads per token, the row stays in registers)
    const int tok = tid >> 2, part = tid & 3;
    const bool tvalid = tok < ntok;
    const float* row = x + ((size_t)b * HW + n0 + (tvalid ? tok : 0)) * C + part * (C / 4);
    float4 xv[QPT];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < QPT; ++j) { xv[j] = reinterpret_cast<const float4*>(row)[j]; s += xv[j].x + xv[j].y + xv[j].z + xv[j].w; }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    const float mean = s / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        const float a0 = xv[j].x - mean, a1 = xv[j].y - mean, a2 = xv[j].z - mean, a3 = xv[j].w - mean;
        q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    }
    q += __shfl_xor(q, 1); q += __shfl_xor(q, 2);
    const float rstd = 1.f / sqrtf(q / C + 1e-5f);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int sl = 0; sl < NS; ++sl) {
        if (sl > 0) __syncthreads();
        // stage the K slice of the normalised tile: the thread's C/4 channels start at part*C/4
        {
            constexpr int CPT = C / 4;                         // channels per thread
#pragma unroll
            for (int g8 = 0; g8 < CPT / 8; ++g8) {
                const int c0 = part * CPT + g8 * 8;            // global channel of this group of 8
                if (c0 / CK == sl) {
                    const float4 a = xv[2 * g8], bq = xv[2 * g8 + 1];
                    float v[8] = {(a.x - mean) * rstd, (a.y - mean) * rstd, (a.z - mean) * rstd, (a.w - mean) * rstd,
                                  (bq.x - mean) * rstd, (bq.y - mean) * rstd, (bq.z - mean) * rstd, (bq.w - mean) * rstd};
                    uint4 hi, lo;
                    split8v(v, hi, lo);
                    if (!tvalid) { hi = make_uint4(0, 0, 0, 0); lo = hi; }
                    char* d = Xs + tok * PITCH + (c0 - sl * CK) * 2;
                    *reinterpret_cast<uint4*>(d) = hi;
                    *reinterpret_cast<uint4*>(d + 2 * CK) = lo;
                }
            }
        }
        // weight slab of this slice: 256 rows x 128 B
        {
            const char* src = reinterpret_cast<const char*>(wkv) + (size_t)sl * 256 * ROWB;
#pragma unroll
            for (int j = 0; j < (256 * (CK / 4)) / 256; ++j) {
                const int idx = tid + j * 256;
                const int r = idx / (CK / 4), piece = idx - r * (CK / 4);
                *reinterpret_cast<uint4*>(Ws + r * PITCH + piece * 16) = *reinterpret_cast<const uint4*>(src + (size_t)r * ROWB + piece * 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < CK / 16; ++ks) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const char* a = Xs + (tm * 32 + l31) * PITCH + half * 16 + ks * 32;
                ah[tm] = *reinterpret_cast<const bf16x8*>(a);
                al[tm] = *reinterpret_cast<const bf16x8*>(a + 2 * CK);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const char* bp = Ws + (w * 64 + tn * 32 + l31) * PITCH + half * 16 + ks * 32;
                bh[tn] = *reinterpret_cast<const bf16x8*>(bp);
                bl[tn] = *reinterpret_cast<const bf16x8*>(bp + 2 * CK);
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                }
        }
    }

    // ---- column softmax statistics of k over the chunk's tokens (rows = registers), p = exp(k - max)
    // Tokens past the end of the map (the last chunk of a map whose size is no multiple of 64): their operand rows were staged as zeros, so
    // their v is exactly 0 already; their k is sent to -3e38 here -- out of the maximum, exp() = 0 -- in a block the full chunks jump over
    // (a per-element select in the loops below cost ~160 of the ~1400 instructions a chunk issues).
    if (ntok < KV_TOK) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= ntok) acc[tm][0][r] = -3.0e38f;
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[tm][0][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float ps = 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pe = __expf(acc[tm][0][r] - mx);
            acc[tm][0][r] = pe;
            ps += pe;
        }
    ps += __shfl_xor(ps, 32);

    // ---- context[d][e] = sum_tok p[tok][d] * v[tok][e]: both tiles feed the MFMA straight from the accumulators
    f32x16 ctx;
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ph, pl, vh, vl;
            acc_to_frag(acc[tm][0], s2, ph, pl);
            acc_to_frag(acc[tm][1], s2, vh, vl);
            ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl, vh, ctx, 0, 0, 0);
            ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, vl, ctx, 0, 0, 0);
            ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, vh, ctx, 0, 0, 0);
        }

    const size_t slot = ((size_t)b * heads + w) * nsplit + sp;
    if (half == 0) { pmax[slot * D + l31] = mx; psum[slot * D + l31] = ps; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * half;      // C/D layout: row = d, column (lane) = e
        pctx[(slot * D + d) * D + l31] = ctx[r];
    }
}

// ---- 64-channel maps: the same computation with the weights RESIDENT in LDS and several 64-token chunks per workgroup --------------
// The kernel above reloads its 64 KB weight image for every 16 KB of activations (one chunk per workgroup): at C = 64 four fifths of its
// memory traffic are weights from L2 and every workgroup is a chain of exposed latencies (x, LayerNorm, weights, MFMA) -- 173 us per launch
// at 256 x 64 x 64, 1.5 TB/s.  Here a workgroup loads the image once (LDS-DMA, both 32-channel slices, unpadded rows with XOR-swizzled
// 16-byte pieces: 64 KB), then walks up to CPW chunks of its sample: the next chunk's token rows are requested while the current chunk
// is multiplied, a chunk's two slices are staged together (two barriers per chunk).  MERGE = false (round 3): each chunk emits the same
// flash-style partials as the kernel above (no state is carried from chunk to chunk; the combine kernel and every result bit as they were).
// LDS: 64 KB weights + 16 KB token tiles = half a CU: two workgroups per CU.
// MERGE (round 4, default): the workgroup carries running column maxima, exp-sums and ONE context across its chunks (online softmax: the
// exponentials of a chunk are taken against max(running, chunk) and only the running context is rescaled -- its rows sit on the register
// index, so the row factors come from their lanes by ds_bpermute) and writes one partial per GROUP of cpw chunks: at 256 x 64 x 64 the
// per-chunk form wrote 268 MB of partials, as much as it read, and linattn_combine read them back.  cpw is a function of the map size
// only (kv64_cpw), never of the batch, so a tile's bits do not depend on what it is batched with.
template <bool MERGE>
__global__ __launch_bounds__(256, 2) void linattn_kv64_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wkv, int HW, int nsplit,
                                                              int cpw, float* __restrict__ pmax, float* __restrict__ psum, float* __restrict__ pctx) {
    constexpr int C = 64, CK = 32, ROWB = 4 * CK, heads = 4, D = 32;
    // (Tried: starting the workgroups 256 .. 511, 768 .. of the grid -- the second one of each CU -- 1 to 5 x 1024 cycles late, so that their VALU
    // phases meet the first one's MFMA phases: 118.8 us without, 120.2 - 124.9 with; profiles/r04_s4_linattn_kv64_stagger_negative.txt.)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ws = smem;                                   // [2 slices][256 rows][128 B]
    char* Xs = smem + 2 * 256 * ROWB;                  // [2 slices][64 tokens][128 B]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int groups = (nsplit + cpw - 1) / cpw;
    const int b = blockIdx.x / groups, sp0 = (blockIdx.x % groups) * cpw, sp1 = min(nsplit, sp0 + cpw);

    // weights: 64 wave-instructions of 1 KiB (8 rows each); LDS takes them linearly, so lane l (slot l & 7 of row l >> 3) fetches piece
    // (l & 7) ^ ((row >> 1) & 7) of its row
    {
        const int wv = __builtin_amdgcn_readfirstlane(w);
        const int r8 = lane >> 3;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int blk = i * 4 + wv;                // 1 KiB block = rows 8 blk .. 8 blk + 7 of the [512][128 B] image
            const int row = blk * 8 + r8;
            const char* src = reinterpret_cast<const char*>(wkv) + (size_t)row * ROWB + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(Ws + blk * 1024), 16, 0, 0);
        }
    }

    const int tok = tid >> 2, part = tid & 3;          // 4 threads per token, 16 channels each: parts 0, 1 -> slice 0, parts 2, 3 -> slice 1
    float4 xv[4];
    auto x_req = [&](int sp) {
        const int n0 = sp * KV_TOK, ntok = min(KV_TOK, HW - n0);
        const float* row = x + ((size_t)b * HW + n0 + (tok < ntok ? tok : 0)) * C + part * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = reinterpret_cast<const float4*>(row)[j];
    };
    x_req(sp0);
    // operand addresses (swizzled): A = token rows, B = this head's 64 weight rows
    int aoff[2], boff[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) { const int r = tm * 32 + l31; aoff[tm] = r * ROWB + ((((r >> 1) & 7) ^ half) << 4); }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) { const int r = w * 64 + tn * 32 + l31; boff[tn] = r * ROWB + ((((r >> 1) & 7) ^ half) << 4); }
    const int xdst = (part >> 1) * 64 * ROWB + tok * ROWB;           // this thread's token row in its slice's tile
    const int xsw = (tok >> 1) & 7, xp0 = (part & 1) * 2;            // its two 8-channel groups are pieces xp0, xp0 + 1 (hi) and + 4 (lo)

    float m_run = -3.0e38f, s_run = 0.f;               // MERGE: column d = l31 of this head (both halves hold the same values)
    f32x16 ctx_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx_run[r] = 0.f;
    for (int sp = sp0; sp < sp1; ++sp) {
        const int n0 = sp * KV_TOK, ntok = min(KV_TOK, HW - n0);
        const bool tvalid = tok < ntok;
        // ---- LayerNorm of the token row (in registers), split, stage
        float s = (xv[0].x + xv[0].y + xv[0].z + xv[0].w) + (xv[1].x + xv[1].y + xv[1].z + xv[1].w) + (xv[2].x + xv[2].y + xv[2].z + xv[2].w) +
                  (xv[3].x + xv[3].y + xv[3].z + xv[3].w);
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
        const float mean = s / C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a0 = xv[j].x - mean, a1 = xv[j].y - mean, a2 = xv[j].z - mean, a3 = xv[j].w - mean;
            q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
        }
        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2);
        const float rstd = 1.f / sqrtf(q / C + 1e-5f);
        uint4 hi[2], lo[2];
#pragma unroll
        for (int g8 = 0; g8 < 2; ++g8) {
            const float4 a = xv[2 * g8], bq = xv[2 * g8 + 1];
            float v[8] = {(a.x - mean) * rstd, (a.y - mean) * rstd, (a.z - mean) * rstd, (a.w - mean) * rstd,
                          (bq.x - mean) * rstd, (bq.y - mean) * rstd, (bq.z - mean) * rstd, (bq.w - mean) * rstd};
            split8v(v, hi[g8], lo[g8]);
            if (!tvalid) { hi[g8] = make_uint4(0, 0, 0, 0); lo[g8] = hi[g8]; }
        }
        if (sp + 1 < sp1) x_req(sp + 1);               // the next chunk's rows travel while this one is multiplied
        // Raw barriers in this loop: the waves only share LDS.  __syncthreads() is also a fence for global memory -- hipcc puts vmcnt(0) in
        // front of it, i.e. every chunk would wait for the round trip of the previous chunk's partial stores and, behind them in the queue,
        // for the rows just requested (measured: 5.6 us per chunk instead of ~1).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the previous chunk's operand reads are done
#pragma unroll
        for (int g8 = 0; g8 < 2; ++g8) {
            *reinterpret_cast<uint4*>(Xs + xdst + (((xp0 + g8) ^ xsw) << 4)) = hi[g8];
            *reinterpret_cast<uint4*>(Xs + xdst + (((xp0 + g8 + 4) ^ xsw) << 4)) = lo[g8];
        }
        if (sp == sp0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first chunk: this wave's weight DMAs have landed (the barrier below publishes them)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ah[2], al[2], bh[2], bl[2];
                const int po = ks * 32;                // pieces 2 ks, 2 ks + 1 (hi); + 64 bytes = lo
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const char* a = Xs + sl * 64 * ROWB;
                    ah[tm] = *reinterpret_cast<const bf16x8*>(a + (aoff[tm] ^ po));
                    al[tm] = *reinterpret_cast<const bf16x8*>(a + (aoff[tm] ^ po ^ 64));
                }
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const char* bp = Ws + sl * 256 * ROWB;
                    bh[tn] = *reinterpret_cast<const bf16x8*>(bp + (boff[tn] ^ po));
                    bl[tn] = *reinterpret_cast<const bf16x8*>(bp + (boff[tn] ^ po ^ 64));
                }
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    }
            }

        // ---- column softmax statistics of k over the chunk's tokens (rows = registers), p = exp(k - max)
        if (ntok < KV_TOK) {                     // tokens past the end of the map: see linattn_kv_fused_kernel
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= ntok) acc[tm][0][r] = -3.0e38f;
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[tm][0][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if constexpr (MERGE) {
            // the running state moves to the new maxima: exp(m_run - m_new) is 0 on the first chunk (m_run = -3e38) and 1 where the maximum stays
            const float m_new = fmaxf(m_run, mx);
            const float f_old = __expf(m_run - m_new);
            s_run *= f_old;
#pragma unroll
            for (int r = 0; r < 16; ++r) ctx_run[r] *= __shfl(f_old, (r & 3) + 8 * (r >> 2) + 4 * half);     // row d of the context <- lane d
            m_run = m_new; mx = m_new;
        }
        float ps = 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pe = __expf(acc[tm][0][r] - mx);
                acc[tm][0][r] = pe;
                ps += pe;
            }
        ps += __shfl_xor(ps, 32);
        f32x16 ctx;
        if constexpr (MERGE) { ctx = ctx_run; s_run += ps; }
        else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ph, pl, vh, vl;
                acc_to_frag(acc[tm][0], s2, ph, pl);
                acc_to_frag(acc[tm][1], s2, vh, vl);
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl, vh, ctx, 0, 0, 0);
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, vl, ctx, 0, 0, 0);
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, vh, ctx, 0, 0, 0);
            }
        if constexpr (MERGE) {
            ctx_run = ctx;
            if (sp + 1 < sp1) continue;                              // one partial per group, after its last chunk
            ps = s_run;
        }
        const size_t slot = MERGE ? ((size_t)b * heads + w) * groups + blockIdx.x % groups : ((size_t)b * heads + w) * nsplit + sp;
        if (half == 0) { pmax[slot * D + l31] = mx; psum[slot * D + l31] = ps; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * half;      // C/D layout: row = d, column (lane) = e
            pctx[(slot * D + d) * D + l31] = ctx[r];
        }
    }
}

// to_qkv weight [384][C] (torch) + LayerNorm gain g[C] -> the kernel's split k/v image
__global__ __launch_bounds__(256) void pack_kv_kernel(const float* __restrict__ wqkv, const float* __restrict__ g, int C,
                                                      unsigned short* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;        // over 256 rows x C
    if (i >= 256 * C) return;
    const int r = i / C, c = i - r * C;
    const int h = r / 64, j = r - h * 64;
    const int src_row = j < 32 ? 128 + h * 32 + j : 256 + h * 32 + (j - 32);   // k block, v block of to_qkv
    const float v = wqkv[(size_t)src_row * C + c] * g[c];
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    const int sl = c / 32, kl = c - sl * 32;
    unsigned short* row = dst + ((size_t)sl * 256 + r) * 64;
    row[kl] = __builtin_bit_cast(unsigned short, hi);
    row[32 + kl] = __builtin_bit_cast(unsigned short, lo);
}

int launch_pack_kv(const float* wqkv, const float* g, int C, unsigned short* dst, hipStream_t st) {
    hipLaunchKernelGGL(pack_kv_kernel, dim3((256 * C + 255) / 256), dim3(256), 0, st, wqkv, g, C, dst);
    return 0;
}

int linattn_kv_nsplit(int HW) { return (HW + KV_TOK - 1) / KV_TOK; }

static bool kv64_old() { static const bool v = getenv("HICDIFF_KV64_OLD") && atoi(getenv("HICDIFF_KV64_OLD")) != 0; return v; }       // A/B switch of the round-3 kernel
static bool kv64_merge() { static const bool v = !(getenv("HICDIFF_KV64_MERGE") && atoi(getenv("HICDIFF_KV64_MERGE")) == 0); return v; }  // 0: one partial per chunk (the round-3 / early round-4 form)

// chunks per workgroup of the merging 64-channel kernel: about eight groups per sample, at most 16 chunks each -- by the MAP SIZE only
// (what a group covers decides the order of the sums, so it must not follow the batch)
static int kv64_cpw(int HW) {
    const int nsplit = linattn_kv_nsplit(HW);
    return std::min(16, std::max(1, (nsplit + 7) / 8));
}

// partials per (sample, head) that launch_linattn_kv_fused writes and linattn_combine reads
int linattn_kv_nparts(int HW, int C) {
    const int nsplit = linattn_kv_nsplit(HW);
    if (C == 64 && !kv64_old() && kv64_merge()) { const int cpw = kv64_cpw(HW); return (nsplit + cpw - 1) / cpw; }
    return nsplit;
}

int launch_linattn_kv_fused(const float* x, const unsigned short* wkv, int B, int HW, int C, float* pmax, float* psum, float* pctx,
                            hipStream_t st) {
    const int nsplit = linattn_kv_nsplit(HW);
    const dim3 grid((unsigned)(B * nsplit));
    if (C == 64 && !kv64_old()) {
        const bool merge = kv64_merge();
        const void* fn = merge ? reinterpret_cast<const void*>(linattn_kv64_kernel<true>) : reinterpret_cast<const void*>(linattn_kv64_kernel<false>);
        if (!hd_raise_dynamic_lds(fn, 80 * 1024)) {
            hd_set_error("linattn_kv64: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); return -3;
        }
        if (merge) {
            const int cpw = kv64_cpw(HW), groups = (nsplit + cpw - 1) / cpw;
            hipLaunchKernelGGL(linattn_kv64_kernel<true>, dim3((unsigned)(B * groups)), dim3(256), 80 * 1024, st, x, wkv, HW, nsplit, cpw, pmax, psum, pctx);
        } else {
            // chunks per workgroup: as many as leave about two rounds of workgroups on 256 CUs x 2 slots, at most 16 (by the map size and the
            // batch only in the grid shape -- the per-chunk partials do not depend on how chunks are grouped)
            int cpw = 16;
            while (cpw > 1 && (long long)B * ((nsplit + cpw - 1) / cpw) < 1024) cpw >>= 1;
            const int groups = (nsplit + cpw - 1) / cpw;
            hipLaunchKernelGGL(linattn_kv64_kernel<false>, dim3((unsigned)(B * groups)), dim3(256), 80 * 1024, st, x, wkv, HW, nsplit, cpw, pmax, psum, pctx);
        }
    } else if (C == 64) hipLaunchKernelGGL(linattn_kv_fused_kernel<64>, grid, dim3(256), 0, st, x, wkv, HW, nsplit, pmax, psum, pctx);
    else if (C == 128) hipLaunchKernelGGL(linattn_kv_fused_kernel<128>, grid, dim3(256), 0, st, x, wkv, HW, nsplit, pmax, psum, pctx);
    else if (C == 256) hipLaunchKernelGGL(linattn_kv_fused_kernel<256>, grid, dim3(256), 0, st, x, wkv, HW, nsplit, pmax, psum, pctx);
    else { hd_set_error("linattn_kv_fused: 64-, 128- and 256-channel maps only"); return -1; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("linattn_kv_fused: ") + hipGetErrorString(e)); return -3; }
    return 0;
}

// ---- q side: the context folded into to_out ----------------------------------------------------------------
// out[n][h*32+e] = scale * sum_d ctx[h][d][e] * softmax_d(q)[n][h*32+d] followed by the 1x1 to_out convolution is one
// 1x1 convolution of softmax_d(q) with the per-sample weight  W'_b[o][h*32+d] = scale * sum_e Wout[o][h*32+e] * ctx_b[h][d][e]
// (src/hicdiff.py:217-226).  This kernel writes W'_b in the split-bf16 image layout of the conv kernel
// ([16-channel k-step = 2 * head + d / 16][CoutPad][16 hi | 16 lo]); the conv's loader then applies the softmax (IN_SOFTMAX32).
// permute != 0: the chained q kernel's own layout instead ([head][CoutPad][32 hi | 32 lo], d in accumulator order).
// grid (B * heads, CoutPad / 64): a workgroup folds one head's context into 64 output rows.  The head's weight slice [32 e][64 o] and the
// context sit in LDS; thread (o = tid & 63, dq = tid >> 6) accumulates d = 8 dq .. 8 dq + 7 (a wave shares its d's: the context reads are
// broadcasts); the split results are assembled as the final byte image in LDS and leave with coalesced 16-byte stores.  (The first version --
// one workgroup per head, weights read from global inside the dot products, 2-byte stores -- took 36 / 69 / 129 us at C = 64 / 128 / 256.)
__global__ __launch_bounds__(256) void linattn_fold_out_kernel(const float* __restrict__ wout, const float* __restrict__ ctx, int CoutPad,
                                                               unsigned short* __restrict__ dst, int permute) {
    constexpr int D = 32;
    __shared__ __attribute__((aligned(16))) float cst[D][D];                 // the context TRANSPOSED, [e][d]: a thread's eight d are two 16-byte broadcast reads
    __shared__ float ws[D][64];
    __shared__ __attribute__((aligned(16))) unsigned short img[64 * 64];     // 8 KB: permute ? [64 o][32 hi | 32 lo] : [2 k-steps][64 o][16 hi | 16 lo]
    const int bh = blockIdx.x, h = bh & 3, o0 = blockIdx.y * 64, tid = threadIdx.x;       // heads = 4
    // twelve loads per thread, all issued before the first LDS write (constant trip counts: as `for (i = tid; i < N; i += 256)` loops hipcc kept
    // them rolled, one L2 round trip per iteration)
    {
        float cr[4], wr[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = tid + k * 256; cr[k] = ctx[(size_t)bh * D * D + (i % D) * D + i / D]; }   // lanes along d: conflict-free LDS writes, 4 KB of strided reads from L2
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = tid + k * 256; wr[k] = wout[(size_t)(h * D + (i >> 6)) * CoutPad + o0 + (i & 63)]; }   // packed fp32 [Cin = 128][CoutPad]
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = tid + k * 256; cst[i / D][i % D] = cr[k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = tid + k * 256; ws[i >> 6][i & 63] = wr[k]; }
    }
    __syncthreads();
    const int o = tid & 63, dq = tid >> 6;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 8
    for (int e = 0; e < D; ++e) {
        const float wv = ws[e][o];
        const float4 c0 = *reinterpret_cast<const float4*>(&cst[e][dq * 8]), c1 = *reinterpret_cast<const float4*>(&cst[e][dq * 8 + 4]);
        acc[0] += wv * c0.x; acc[1] += wv * c0.y; acc[2] += wv * c0.z; acc[3] += wv * c0.w;
        acc[4] += wv * c1.x; acc[5] += wv * c1.y; acc[6] += wv * c1.z; acc[7] += wv * c1.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = dq * 8 + j;
        const float v = acc[j] * 0.17677669529663687f;               // dim_head ** -0.5
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)hi);
        if (permute) {
            // d axis in the order an MFMA accumulator presents it as an operand (linattn_q_fused.hip): position 16*s + 8*half + 4*g + i  <-  d = 16*s + 8*g + 4*half + i
            const int pos = (d & 16) | ((d & 4) << 1) | ((d & 8) >> 1) | (d & 3);
            img[o * 64 + pos] = __builtin_bit_cast(unsigned short, hi);
            img[o * 64 + D + pos] = __builtin_bit_cast(unsigned short, lo);
        } else {
            unsigned short* row = img + ((d >> 4) * 64 + o) * 32;
            row[d & 15] = __builtin_bit_cast(unsigned short, hi);
            row[16 + (d & 15)] = __builtin_bit_cast(unsigned short, lo);
        }
    }
    __syncthreads();
    // 8 KB out: permute: rows o0 .. o0+63 of [b][head][CoutPad][64 shorts] (contiguous); else two 4 KB runs, k-steps 2 h and 2 h + 1 of [b][8][CoutPad][32 shorts]
    const uint4* src = reinterpret_cast<const uint4*>(img);
    char* out = reinterpret_cast<char*>(dst) + (size_t)bh * CoutPad * 128;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + k * 256;                                 // 16-byte piece of the image
        char* g = permute ? out + (size_t)o0 * 128 + i * 16 : out + (size_t)(i >> 8) * CoutPad * 64 + (size_t)o0 * 64 + (i & 255) * 16;
        *reinterpret_cast<uint4*>(g) = src[i];
    }
}

int launch_linattn_fold_out(const float* wout_packed, const float* ctx, int B, int CoutPad, unsigned short* dst, hipStream_t st, int permute) {
    hipLaunchKernelGGL(linattn_fold_out_kernel, dim3(B * 4, CoutPad / 64), dim3(256), 0, st, wout_packed, ctx, CoutPad, dst, permute);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("linattn_fold_out launch: ") + hipGetErrorString(e)); return -3; }
    return 0;
}
