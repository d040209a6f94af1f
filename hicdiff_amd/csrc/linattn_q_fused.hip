// Chained query side of LinearAttention for 64-channel feature maps (src/hicdiff.py:212-226, 207-210, 64-70):
//
//   out = LayerNorm_c( to_out( einsum(context, softmax_d(to_q(LayerNorm_c(x))) * scale) ) ) * g_out + x
//
// in ONE kernel: neither q (128 channels, twice the size of x) nor the attention output nor the pre-LayerNorm tensor
// reaches HBM -- the tensor round trips, not the arithmetic, are what this side of the block costs.
//   * the per-sample context is already folded into to_out's weight by linattn_fold_out_kernel
//     (W'_b = scale * Wout . ctx_b, split-bf16 slabs [head][64 rows][32 hi | 32 lo]);
//   * to_q's weight comes with the PreNorm gain folded in, split the same way ([C/32 slices][128 rows][32 hi | 32 lo]);
//   * the PreNorm statistics (mean, rstd per pixel) come from the block that produced x.
// One workgroup of 8 waves per 256 pixels of one sample (two workgroups per CU); a wave owns 32 pixels end to end, so
// between the barrier that publishes the two weight images and the one before the epilogue no wave waits for another:
//   x row -> LayerNorm -> split-bf16 A fragments in REGISTERS (each lane loads the 8 channels its fragment needs);
//   per head: q_h^T = Wq_h . A^T (12 MFMAs; the transposed product puts the head's 32 channels on the REGISTER index, so
//   the softmax is an in-lane reduction plus one exchange with lane ^ 32), p goes from the accumulator registers straight
//   into the operand of y += p . W'_b,h (12 MFMAs; W' is stored with its d axis in the accumulator's order);
//   epilogue through the wave's LDS stage: bias, channel LayerNorm over the 64 outputs, gain, + x, 16-byte stores.
#include "hd_common.h"

#include <set>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int QF_PITCH = 144;                 // one 32-channel split row: 32 bf16 hi | 32 bf16 lo | 16 B pad (odd multiple of 16)
constexpr int QF_WQ_BYTES = 2 * 128 * QF_PITCH;   // [2 slices][128 q rows]
constexpr int QF_WF_BYTES = 4 * 64 * QF_PITCH;    // [4 heads][64 output rows]
constexpr int QF_STAGE = 32 * 68 * 4;             // per wave: epilogue staging [32 px][64 + 4] floats
static_assert(8 * QF_STAGE <= QF_WQ_BYTES + QF_WF_BYTES, "the epilogue staging overlays the two weight images");
constexpr int QF_LDS = QF_WQ_BYTES + QF_WF_BYTES;   // 73,728 B: two workgroups (16 waves) per CU

template <int CTRL>
__device__ __forceinline__ float dppf(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
// all-reduce over a 16-lane DPP row: quad butterflies, half-row and row mirrors
__device__ __forceinline__ float allsum16(float x) {
    x += dppf<0xB1>(x); x += dppf<0x4E>(x); x += dppf<0x141>(x); x += dppf<0x140>(x);
    return x;
}

__device__ __forceinline__ void split8r(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 hh = (__bf16)v[j];
        hi[j] = hh;
        lo[j] = (__bf16)(v[j] - (float)hh);
    }
}

__global__ __launch_bounds__(512) void linattn_q_fused_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                              const unsigned short* __restrict__ wq, const unsigned short* __restrict__ wfold,
                                                              const float* __restrict__ bias, const float* __restrict__ gout,
                                                              float* __restrict__ out, int HW, int tiles_per_sample) {
    constexpr int C = 64, D = 32;
    extern __shared__ __attribute__((aligned(16))) char qf_lds[];
    char* WQs = qf_lds;
    char* WFs = WQs + QF_WQ_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    char* stg = qf_lds + w * QF_STAGE;                  // epilogue staging: overlays the weight images once every wave is done with them
    const int b = blockIdx.x / tiles_per_sample, tile = blockIdx.x % tiles_per_sample;
    const int n0 = tile * 256 + w * 32;                 // first pixel of this wave inside the sample

    // ---- this lane's pixel: 32 of its 64 channels (the A fragments of the four k16 steps), LayerNorm, split
    const int n = n0 + l31;
    const bool valid = n < HW;
    const size_t prow = (size_t)b * HW + (valid ? n : 0);
    float4 raw[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float* g = x + prow * C + ks * 16 + half * 8;
        raw[ks][0] = *reinterpret_cast<const float4*>(g);
        raw[ks][1] = *reinterpret_cast<const float4*>(g + 4);
    }
    const float2 st = *reinterpret_cast<const float2*>(stats + 2 * prow);
    float mu, rs;   // copied out of the load pair (conv_bf16x3_kernel.h: packed multiplies broadcasting from an odd register)
    asm volatile("v_mov_b32 %0, %1" : "=v"(mu) : "v"(st.x));
    asm volatile("v_mov_b32 %0, %1" : "=v"(rs) : "v"(st.y));

    // ---- the two weight images -> LDS (16 bytes per lane; rows are 128 B of data + pad)
    // Both images are 2048 pieces of 16 bytes = four per thread each: all eight loads are issued before the first LDS write.  (As two
    // `for (i = tid; i < 2048; i += 512)` loops hipcc kept them rolled -- load, wait, write, four times each: eight L2 round trips in a row
    // in front of the first barrier of every workgroup.)
    {
        const char* wf = reinterpret_cast<const char*>(wfold) + (size_t)b * (4 * 64 * 128);
        uint4 wreg[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + k * 512, row = i >> 3, piece = i & 7;
            wreg[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wq) + (size_t)row * 128 + piece * 16);
            wreg[4 + k] = *reinterpret_cast<const uint4*>(wf + (size_t)row * 128 + piece * 16);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + k * 512, row = i >> 3, piece = i & 7;
            *reinterpret_cast<uint4*>(WQs + row * QF_PITCH + piece * 16) = wreg[k];
            *reinterpret_cast<uint4*>(WFs + row * QF_PITCH + piece * 16) = wreg[4 + k];
        }
    }
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float v[8] = {(raw[ks][0].x - mu) * rs, (raw[ks][0].y - mu) * rs, (raw[ks][0].z - mu) * rs, (raw[ks][0].w - mu) * rs,
                      (raw[ks][1].x - mu) * rs, (raw[ks][1].y - mu) * rs, (raw[ks][1].z - mu) * rs, (raw[ks][1].w - mu) * rs};
        if (!valid) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        split8r(v, ah[ks], al[ks]);
    }
    __syncthreads();                      // weight images visible; from here on the waves are independent

    f32x16 y[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[tn][r] = 0.f;

#pragma unroll 1
    for (int h = 0; h < 4; ++h) {
        // q_h^T[d][px] = Wq[h*32 + d][:] . LN(x)[px][:]: the TRANSPOSED product (operands swapped; the fragments are the
        // same registers), so that d sits on the register index: row (r&3) + 8*(r>>2) + 4*half, column (lane) = pixel.
        f32x16 q;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const char* bp = WQs + ((ks >> 1) * 128 + h * D + l31) * QF_PITCH + (ks & 1) * 32 + half * 16;
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(bp);
            const bf16x8 wl = *reinterpret_cast<const bf16x8*>(bp + 64);
            q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah[ks], q, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al[ks], q, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah[ks], q, 0, 0, 0);
        }
        // softmax over d: 16 values in this lane, the other 16 in lane ^ 32
        float m = q[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, q[r]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { q[r] = __expf(q[r] - m); sum += q[r]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] *= inv;
        // y[px][o] += sum_d p[px][d] . W'_b[h][o][d].  p is used straight from the accumulator registers: registers 8*s..8*s+7 of
        // lane (px, half) hold d = 16*s + 4*half + {0..3} and 16*s + 8 + 4*half + {0..3}; linattn_fold_out stored W' with its d
        // axis in exactly that order, so both operands agree on the (permuted) contraction order and nothing moves between lanes.
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = q[8 * s2 + j];
                const __bf16 hh = (__bf16)v;
                ph[j] = hh;
                pl[j] = (__bf16)(v - (float)hh);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const char* bp = WFs + (h * 64 + tn * 32 + l31) * QF_PITCH + s2 * 32 + half * 16;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + 64);
                y[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl, bh, y[tn], 0, 0, 0);
                y[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, bl, y[tn], 0, 0, 0);
                y[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, bh, y[tn], 0, 0, 0);
            }
        }
    }

    // ---- epilogue through the wave's stage [32 px][68 floats]: rows become contiguous, 16 lanes per row
    // The residual rows (x again, in the epilogue's lane layout: mostly L2 / MALL hits) are requested BEFORE the barrier, so that their
    // round trip runs under the wait for the other waves and the staging instead of in front of the first row pass.
    const int cq = lane & 15, rg = lane >> 4;           // 4 channels, rows rg, rg + 4, ...
    float4 rr[8];
#ifndef HD_QF_LATE_RESIDUAL   // A/B builds (make TAG=_qlate EXTRA=-DHD_QF_LATE_RESIDUAL): the form before, one exposed load per row pass
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int nn = n0 + pass * 4 + rg;
        rr[pass] = *reinterpret_cast<const float4*>(x + ((size_t)b * HW + (nn < HW ? nn : 0)) * C + cq * 4);
    }
#endif
    __syncthreads();                      // the stage overlays the weight images: every wave has left the head loop
    float* sf = reinterpret_cast<float*>(stg);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) sf[((r & 3) + 8 * (r >> 2) + 4 * half) * 68 + tn * 32 + l31] = y[tn][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g4 = *reinterpret_cast<const float4*>(gout + cq * 4);
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 4 + rg, nn = n0 + row;
        const size_t o = ((size_t)b * HW + (nn < HW ? nn : 0)) * C + cq * 4;
#ifdef HD_QF_LATE_RESIDUAL
        rr[pass] = *reinterpret_cast<const float4*>(x + o);
#endif
        const float4 a4 = *reinterpret_cast<const float4*>(sf + row * 68 + cq * 4);
        float v0 = a4.x + b4.x, v1 = a4.y + b4.y, v2 = a4.z + b4.z, v3 = a4.w + b4.w;
        const float mean = allsum16((v0 + v1) + (v2 + v3)) * (1.f / C);
        v0 -= mean; v1 -= mean; v2 -= mean; v3 -= mean;
        const float var = allsum16(v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3) * (1.f / C);
        const float r2 = rsqrtf(var + 1e-5f);
        const f32x4 o4 = {v0 * r2 * g4.x + rr[pass].x, v1 * r2 * g4.y + rr[pass].y, v2 * r2 * g4.z + rr[pass].z, v3 * r2 * g4.w + rr[pass].w};
        if (nn < HW) *reinterpret_cast<f32x4*>(out + o) = o4;
    }
}

// to_qkv weight [384][C] (torch) rows 0..127 with the PreNorm gain folded in -> [C/32 slices][128 rows][32 hi | 32 lo]
__global__ __launch_bounds__(256) void pack_q_kernel(const float* __restrict__ wqkv, const float* __restrict__ g, int C,
                                                     unsigned short* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;        // over 128 rows x C
    if (i >= 128 * C) return;
    const int r = i / C, c = i - r * C;
    const float v = wqkv[(size_t)r * C + c] * g[c];
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    const int sl = c / 32, kl = c - sl * 32;
    unsigned short* row = dst + ((size_t)sl * 128 + r) * 64;
    row[kl] = __builtin_bit_cast(unsigned short, hi);
    row[32 + kl] = __builtin_bit_cast(unsigned short, lo);
}

}  // namespace

int launch_pack_q(const float* wqkv, const float* g, int C, unsigned short* dst, hipStream_t st) {
    hipLaunchKernelGGL(pack_q_kernel, dim3((128 * C + 255) / 256), dim3(256), 0, st, wqkv, g, C, dst);
    return 0;
}

// x, out: [B][HW][64]; stats: [B*HW][2]; wq: pack_q image; wfold: [B][4][64][64 shorts] (linattn_fold_out, CoutPad = 64)
int launch_linattn_q_fused(const float* x, const float* stats, const unsigned short* wq, const unsigned short* wfold, const float* bias,
                           const float* gout, float* out, int B, int HW, int C, hipStream_t st) {
    if (C != 64) { hd_set_error("linattn_q_fused: 64-channel maps only"); return -1; }
    if (!hd_raise_dynamic_lds(reinterpret_cast<const void*>(linattn_q_fused_kernel), QF_LDS)) {
        hd_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); return -3;
    }
    const int tiles = (HW + 255) / 256;
    hipLaunchKernelGGL(linattn_q_fused_kernel, dim3((unsigned)(B * tiles)), dim3(512), QF_LDS, st, x, stats, wq, wfold, bias, gout, out, HW, tiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("linattn_q_fused launch: ") + hipGetErrorString(e)); return -3; }
    return 0;
}
