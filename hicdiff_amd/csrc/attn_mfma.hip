// Mid-block softmax attention (src/hicdiff.py:229-251) on the matrix cores, for n = H*W <= 64 tokens (8 x 8 at 64 x 64 tiles,
// 5 x 5 at 40 x 40): the one dense contraction the path has besides the convolutions.
//
//   sim[i][j] = scale * q_i . k_j ;  attn = softmax_j(sim) ;  out_i = sum_j attn[i][j] v_j          (4 heads x 32 channels)
//
// One 4-wave workgroup per (sample, head pair); a wave owns one head and 32 queries.  Both products run transposed so that no
// accumulator ever has to change lanes:
//   S^T = K Q^T      : rows = keys (register index), columns = queries (lane)  -> the softmax over keys is an in-lane reduction
//                       over 32 registers plus one exchange with lane ^ 32;
//   O^T = V^T P^T    : P^T (= the softmaxed accumulators, converted to bf16 hi / lo in place) is the B operand as it stands
//                       (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's operand"); V^T is staged in
//                       LDS with its keys in the accumulator's k-order: element j of lane half h of k-step s <-> key
//                       16 s + 8 (j >> 2) + 4 h + (j & 3).
// fp32 operands go through the split-bf16 x3 products like everything else (hi*hi + hi*lo + lo*hi, fp32 accumulate).
#include "hd_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int D = 32, NMAX = 64;
constexpr int QK_PITCH = 2 * D * 2 + 16;        // [32 hi | 32 lo] bf16 + pad: 144 B, an odd multiple of 16
constexpr int VT_PITCH = 2 * NMAX * 2 + 16;     // one d-row of V^T: [64 keys hi | 64 keys lo] + pad: 272 B (17 x 16)

__device__ __forceinline__ void split1(float v, unsigned short& hi, unsigned short& lo) {
    const __bf16 h = (__bf16)v;
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
}

__global__ __launch_bounds__(256) void attn_full_mfma_kernel(const float* __restrict__ qkv, int HW, int heads, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char Qs[2][NMAX * QK_PITCH];
    __shared__ __attribute__((aligned(16))) char Ks[2][NMAX * QK_PITCH];
    __shared__ __attribute__((aligned(16))) char Vt[2][D * VT_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int pairs = heads >> 1;
    const int b = blockIdx.x / pairs, hp = blockIdx.x % pairs;
    const int C3 = 3 * heads * D;                      // 384
    // ---- stage q (pre-scaled), k rows and v transposed, split into bf16 hi / lo; rows >= HW are zero
    for (int e = tid; e < 2 * NMAX * D; e += 256) {    // (head of the pair, token, channel)
        const int hh = e / (NMAX * D), r = e - hh * NMAX * D, tok = r / D, d = r - tok * D;
        const int h = hp * 2 + hh;
        float q = 0.f, k = 0.f, v = 0.f;
        if (tok < HW) {
            const float* src = qkv + ((size_t)b * HW + tok) * C3 + h * D + d;
            q = src[0] * 0.17677669529663687f;          // dim_head ** -0.5
            k = src[heads * D];
            v = src[2 * heads * D];
        }
        unsigned short hi, lo;
        split1(q, hi, lo);
        *reinterpret_cast<unsigned short*>(Qs[hh] + tok * QK_PITCH + d * 2) = hi;
        *reinterpret_cast<unsigned short*>(Qs[hh] + tok * QK_PITCH + 2 * D + d * 2) = lo;
        split1(k, hi, lo);
        *reinterpret_cast<unsigned short*>(Ks[hh] + tok * QK_PITCH + d * 2) = hi;
        *reinterpret_cast<unsigned short*>(Ks[hh] + tok * QK_PITCH + 2 * D + d * 2) = lo;
        // V^T[d][slot(tok)]: keys in the accumulator's k-order
        const int kb = tok >> 5, o = tok & 15, s = (tok >> 4) & 1;
        const int slot = ((kb * 2 + s) * 2 + ((o >> 2) & 1)) * 8 + (o & 3) + 4 * (o >> 3);
        split1(v, hi, lo);
        *reinterpret_cast<unsigned short*>(Vt[hh] + d * VT_PITCH + slot * 2) = hi;
        *reinterpret_cast<unsigned short*>(Vt[hh] + d * VT_PITCH + 2 * NMAX + slot * 2) = lo;
    }
    __syncthreads();
    const int hh = wave >> 1, qb = wave & 1;           // this wave: head hh of the pair, queries qb * 32 .. + 31
    const int h = hp * 2 + hh;
    // ---- S^T[key][query] = K Q^T, two key blocks x two k16 steps x three products
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const char* qa = Qs[hh] + (qb * 32 + l31) * QK_PITCH + ks * 32 + half * 16;
        const bf16x8 qh = *reinterpret_cast<const bf16x8*>(qa), ql = *reinterpret_cast<const bf16x8*>(qa + 2 * D);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const char* ka = Ks[hh] + (kb * 32 + l31) * QK_PITCH + ks * 32 + half * 16;
            const bf16x8 kh = *reinterpret_cast<const bf16x8*>(ka), kl = *reinterpret_cast<const bf16x8*>(ka + 2 * D);
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, st[kb], 0, 0, 0);     // A = keys (rows), B = queries (columns)
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, st[kb], 0, 0, 0);
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, st[kb], 0, 0, 0);
        }
    }
    // ---- softmax over keys for this lane's query: registers (+ the other lane half); keys >= HW are masked out
    float mx = -3.0e38f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key >= HW) st[kb][r] = -3.0e38f;
            mx = fmaxf(mx, st[kb][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float pv = key < HW ? __expf(st[kb][r] - mx) : 0.f;
            st[kb][r] = pv;
            sum += pv;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    // ---- O^T[d][query] = V^T P^T: P^T straight from the accumulators (k-step s of key block kb = registers 8 s .. 8 s + 7)
    f32x16 ot;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pv = st[kb][8 * s + j];
                const __bf16 hj = (__bf16)pv;
                ph[j] = hj;
                pl[j] = (__bf16)(pv - (float)hj);
            }
            const char* va = Vt[hh] + l31 * VT_PITCH + (((kb * 2 + s) * 2 + half) * 8) * 2;
            const bf16x8 vh = *reinterpret_cast<const bf16x8*>(va), vl = *reinterpret_cast<const bf16x8*>(va + 2 * NMAX);
            ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, ot, 0, 0, 0);           // A = V^T (rows d), B = P^T (columns = queries)
            ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, ot, 0, 0, 0);
            ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, ot, 0, 0, 0);
        }
    // ---- out[b][query][h * 32 + d]: this lane holds d = 8 g + 4 half + 0..3 for g = 0..3 of query qb * 32 + l31
    const int query = qb * 32 + l31;
    if (query < HW) {
        float* o = out + ((size_t)b * HW + query) * (heads * D) + h * D + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(o + 8 * g) = make_float4(ot[4 * g] * inv, ot[4 * g + 1] * inv, ot[4 * g + 2] * inv, ot[4 * g + 3] * inv);
    }
}

}  // namespace

// n <= 64 tokens and an even number of heads: the MFMA kernel; anything else stays on the streaming kernel of small_kernels.hip
int launch_attn_full_mfma(const float* qkv, int B, int HW, int heads, float* out, hipStream_t st) {
    if (HW > NMAX || (heads & 1)) return 1;
    hipLaunchKernelGGL(attn_full_mfma_kernel, dim3(B * (heads / 2)), dim3(256), 0, st, qkv, HW, heads, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("attn_full_mfma: ") + hipGetErrorString(e)); return -3; }
    return 0;
}
