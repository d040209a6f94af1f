// DDRM with a non-identity degradation H = U S V^T (src/functions/denoising.py:11-111 with the operator zoo of
// src/functions/svd_replacement.py): the per-step elementwise part in the spectral domain, plus the building blocks the
// SVD-free operators are made of (column gathers, one small matrix applied to many short vectors, the fast
// Walsh-Hadamard transform).  The dense S x S products of the deblurring operators are plain library GEMMs on the host side.
//
// HiCDiff itself only ever selects H = I (hd_ddrm_step fuses that case); these are the rest of SURVEY.md row f-4.
#include "hd_common.h"
#include "../../include/hicdiff_hip.h"

// Philox4x32-10 + Box-Muller, the generator of small_kernels.hip (same key material: seed, tile, step, stream)
__device__ __forceinline__ void philox_round_g(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__device__ __forceinline__ float4 philox_normal4_g(uint64_t seed, uint32_t quad, uint64_t tile, uint32_t step, uint32_t stream) {
    uint32_t c[4] = {quad, (uint32_t)tile, (uint32_t)(tile >> 32) ^ (stream << 28), step};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round_g(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const float u0 = ((float)(c[0] >> 8) + 0.5f) * (1.f / 16777216.f), u1 = ((float)(c[1] >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u2 = ((float)(c[2] >> 8) + 0.5f) * (1.f / 16777216.f), u3 = ((float)(c[3] >> 8) + 0.5f) * (1.f / 16777216.f);
    const float r0 = sqrtf(-2.f * __logf(u0)), r1 = sqrtf(-2.f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

// x0_t = (x_t - eps * sqrt(1 - a_t)) / sqrt(a_t)     (:66)
__global__ __launch_bounds__(256) void ddrm_x0_kernel(const float* __restrict__ x, const float* __restrict__ eps, float sqrt_at, float sqrt_1m_at,
                                                      float* __restrict__ x0, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 xv = reinterpret_cast<const float4*>(x)[i], ev = reinterpret_cast<const float4*>(eps)[i];
    reinterpret_cast<float4*>(x0)[i] = make_float4((xv.x - ev.x * sqrt_1m_at) / sqrt_at, (xv.y - ev.y * sqrt_1m_at) / sqrt_at,
                                                   (xv.z - ev.z * sqrt_1m_at) / sqrt_at, (xv.w - ev.w * sqrt_1m_at) / sqrt_at);
}

// Spectral-domain update of one step (:69-104), element k of sample b (singular value s_k for k < M, none beyond):
//   default ("missing")        V^T x0 + sqrt(sn^2 - (sn etaC)^2) V^T eps + sn etaC z0
//   s_k sn < sigma_0 ("after")  V^T x0 + sqrt(sn^2 - (sn etaA)^2) (U^T y - s_k V^T x0) / sigma_0 + sn etaA z1
//   s_k sn > sigma_0 ("before") (U^T y / s_k) etaB + (1 - etaB) V^T x0 + sqrt(sn^2 - sigma_0^2 / s_k^2 etaB^2) z2
// times sqrt(a_next) (V is linear, so the scale of :104 is applied here).  z0 / z1 / z2: full-layout replayed noise ([B][D], [B][D],
// [B][M]) or NULL for device noise, streams 0 / 1 / 2 keyed by (seed, tile_off + b, step) with the element quad as counter.
__global__ __launch_bounds__(256) void ddrm_general_update_kernel(const float* __restrict__ vt_x0, const float* __restrict__ vt_et,
                                                                  const float* __restrict__ ut_y, const float* __restrict__ sing, int M, int D,
                                                                  const float* __restrict__ z0, const float* __restrict__ z1,
                                                                  const float* __restrict__ z2, float sigma_next, float sigma_0, float etaA,
                                                                  float etaB, float etaC, float sqrt_at_next, float* __restrict__ out, int B,
                                                                  uint64_t seed, uint64_t tile_off, uint32_t step) {
    const int D4 = D >> 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * D4) return;
    const int b = (int)(i / D4), q = (int)(i - (size_t)b * D4);
    const float4 a4 = reinterpret_cast<const float4*>(vt_x0)[i], e4 = reinterpret_cast<const float4*>(vt_et)[i];
    const float av[4] = {a4.x, a4.y, a4.z, a4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
    float n0[4], n1[4], n2[4];
    if (z0) { const float4 t = reinterpret_cast<const float4*>(z0)[i]; n0[0] = t.x; n0[1] = t.y; n0[2] = t.z; n0[3] = t.w; }
    else { const float4 t = philox_normal4_g(seed, q, tile_off + b, step, 0); n0[0] = t.x; n0[1] = t.y; n0[2] = t.z; n0[3] = t.w; }
    if (z1) { const float4 t = reinterpret_cast<const float4*>(z1)[i]; n1[0] = t.x; n1[1] = t.y; n1[2] = t.z; n1[3] = t.w; }
    else { const float4 t = philox_normal4_g(seed, q, tile_off + b, step, 1); n1[0] = t.x; n1[1] = t.y; n1[2] = t.z; n1[3] = t.w; }
    const float4 t2 = z2 ? make_float4(0.f, 0.f, 0.f, 0.f) : philox_normal4_g(seed, q, tile_off + b, step, 2);
    n2[0] = t2.x; n2[1] = t2.y; n2[2] = t2.z; n2[3] = t2.w;
    const float std_c = sigma_next * etaC, til_c = sqrtf(sigma_next * sigma_next - std_c * std_c);
    const float std_a = sigma_next * etaA, til_a = sqrtf(sigma_next * sigma_next - std_a * std_a);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = q * 4 + j;
        float v = av[j] + til_c * ev[j] + std_c * n0[j];
        if (k < M) {
            const float s = sing[k], uy = ut_y[(size_t)b * M + k];
            if (s * sigma_next < sigma_0) v = av[j] + til_a * ((uy - s * av[j]) / sigma_0) + std_a * n1[j];
            if (s * sigma_next > sigma_0) {
                const float zz = z2 ? z2[(size_t)b * M + k] : n2[j];
                v = (uy / s) * etaB + (1.f - etaB) * av[j] + sqrtf(sigma_next * sigma_next - sigma_0 * sigma_0 / (s * s) * (etaB * etaB)) * zz;
            }
        }
        o[j] = sqrt_at_next * v;
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
}

// dst[b][i] = idx[i] >= 0 ? src[b][idx[i]] : 0     (the permutations / selections of Inpainting, SuperResolution, SRConv, CS)
__global__ __launch_bounds__(256) void gather_cols_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int B,
                                                          int Dsrc, int Ddst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * Ddst) return;
    const int b = (int)(i / Ddst), c = (int)(i - (size_t)b * Ddst);
    const int s = idx[c];
    dst[i] = s >= 0 ? src[(size_t)b * Dsrc + s] : 0.f;
}

// dst[n][r] = sum_c mat[r][c] * src[n][c], K <= 64: one small matrix applied to N contiguous K-vectors (the per-patch factor of
// SuperResolution, the per-pixel factor of Colorization)
__global__ __launch_bounds__(256) void kvec_matmul_kernel(const float* __restrict__ src, const float* __restrict__ mat, float* __restrict__ dst,
                                                          size_t N, int K) {
    __shared__ float m[64 * 64];
    for (int i = threadIdx.x; i < K * K; i += 256) m[i] = mat[i];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * K) return;
    const size_t n = i / K;
    const int r = (int)(i - n * K);
    float acc = 0.f;
    for (int c = 0; c < K; ++c) acc += m[r * K + c] * src[n * K + c];
    dst[i] = acc;
}

// dst[i] = A x[i] Bm for n images of S x S (S <= 64, all row-major): the separable operators' V / V^T / U / U^T (Deblurring, Deblurring2D,
// src/functions/svd_replacement.py:401-541: a Kronecker factor on the rows and one on the columns of every channel image).  One workgroup
// per image; A, Bm, the image and the intermediate T = x Bm live in LDS (64 KB at S = 64); fp32 FMA in a fixed order.
__global__ __launch_bounds__(256) void sandwich_matmul_kernel(const float* __restrict__ A, const float* __restrict__ x, const float* __restrict__ Bm,
                                                              float* __restrict__ dst, int S) {
    extern __shared__ float sm[];
    float *As = sm, *Bs = sm + S * S, *Xs = sm + 2 * S * S, *Ts = sm + 3 * S * S;
    const float* xi = x + (size_t)blockIdx.x * S * S;
    for (int i = threadIdx.x; i < S * S; i += 256) { As[i] = A[i]; Bs[i] = Bm[i]; Xs[i] = xi[i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < S * S; i += 256) {
        const int r = i / S, c = i - r * S;
        float acc = 0.f;
        for (int k = 0; k < S; ++k) acc += Xs[r * S + k] * Bs[k * S + c];
        Ts[i] = acc;
    }
    __syncthreads();
    float* o = dst + (size_t)blockIdx.x * S * S;
    for (int i = threadIdx.x; i < S * S; i += 256) {
        const int r = i / S, c = i - r * S;
        float acc = 0.f;
        for (int k = 0; k < S; ++k) acc += As[r * S + k] * Ts[k * S + c];
        o[i] = acc;
    }
}

// dst[n][m] = sum_k src[n][k] mat[k][m] for any sizes: the dense-SVD operator (GeneralH, src/functions/svd_replacement.py:72-107 -- d x d factors,
// memory-hungry upstream too, used by no driver).  64 x 64 output tiles, 16-deep K steps through LDS, 4 x 4 outputs per thread.
__global__ __launch_bounds__(256) void dense_matmul_kernel(const float* __restrict__ src, const float* __restrict__ mat, float* __restrict__ dst,
                                                           int N, int K, int M) {
    __shared__ float sa[16][64 + 1], sb[16][64 + 1];
    const int n0 = blockIdx.y * 64, m0 = blockIdx.x * 64, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 16 * 64; i += 256) {
            const int kk = i & 15, r = i >> 4;                      // src tile: 64 rows x 16 k
            sa[kk][r] = (n0 + r < N && k0 + kk < K) ? src[(size_t)(n0 + r) * K + k0 + kk] : 0.f;
            const int c = i & 63, k2 = i >> 6;                      // mat tile: 16 k x 64 columns
            sb[k2][c] = (k0 + k2 < K && m0 + c < M) ? mat[(size_t)(k0 + k2) * M + m0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = sa[kk][ty * 4 + u]; b[u] = sb[kk][tx * 4 + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
        }
        __syncthreads();
    }
    for (int u = 0; u < 4; ++u)
        for (int v = 0; v < 4; ++v)
            if (n0 + ty * 4 + u < N && m0 + tx * 4 + v < M) dst[(size_t)(n0 + ty * 4 + u) * M + m0 + tx * 4 + v] = acc[u][v];
}

// In-place fast Walsh-Hadamard transform of N rows of length L = 2^p (<= 4096), scaled by `scale` (the reference divides by img_dim:
// src/functions/svd_replacement.py:287-297); one workgroup per row, the row lives in LDS.
__global__ __launch_bounds__(256) void fwht_kernel(float* __restrict__ data, int L, float scale) {
    extern __shared__ float row[];
    float* g = data + (size_t)blockIdx.x * L;
    for (int i = threadIdx.x; i < L; i += 256) row[i] = g[i];
    __syncthreads();
    for (int h = 1; h < L; h <<= 1) {
        for (int i = threadIdx.x; i < L / 2; i += 256) {
            const int blk = i / h, off = i - blk * h, a = blk * 2 * h + off;
            const float x = row[a], y = row[a + h];
            row[a] = x + y; row[a + h] = x - y;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < L; i += 256) g[i] = row[i] * scale;
}

static int check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string(what) + ": " + hipGetErrorString(e)); return HD_EHIP; }
    return HD_OK;
}

extern "C" {

int hd_ddrm_x0(const float* x, const float* eps, float sqrt_at, float sqrt_1m_at, float* x0_out, size_t n, void* stream) {
    if (!x || !eps || !x0_out || n % 4) return HD_EINVAL;
    if (n == 0) return HD_OK;
    hipLaunchKernelGGL(ddrm_x0_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, eps, sqrt_at, sqrt_1m_at, x0_out, n / 4);
    return check("ddrm_x0");
}

int hd_ddrm_general_update(const float* vt_x0, const float* vt_et, const float* ut_y, const float* singulars, int M, const float* z0,
                           const float* z1, const float* z2, const hd_ddrm_coef* cin, float* out, int B, int D, uint64_t seed,
                           uint64_t tile_offset, uint32_t step, void* stream) {
    if (!vt_x0 || !vt_et || !out || !cin || B < 1 || D < 4 || D % 4 || M < 0 || M > D || (M > 0 && (!ut_y || !singulars))) return HD_EINVAL;
    hd_ddrm_coef kk;                      // size-prefixed struct (hicdiff_hip.h): the same rule as hd_ddrm_step
    if (!hd_read_prefixed(cin, sizeof(hd_ddrm_coef), &kk)) { hd_set_error("hd_ddrm_coef: struct_bytes is not a size this library knows"); return HD_EINVAL; }
    const hd_ddrm_coef* c = &kk;
    hipLaunchKernelGGL(ddrm_general_update_kernel, dim3((unsigned)(((size_t)B * (D / 4) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vt_x0, vt_et,
                       ut_y, singulars, M, D, z0, z1, z2, c->sigma_next, c->sigma_0, c->etaA, c->etaB, c->etaC, c->sqrt_at_next, out, B, seed,
                       tile_offset, step);
    return check("ddrm_general_update");
}

int hd_gather_cols(const float* src, const int* idx, float* dst, int B, int Dsrc, int Ddst, void* stream) {
    if (!src || !idx || !dst || B < 0 || Dsrc < 1 || Ddst < 0) return HD_EINVAL;
    if (B == 0 || Ddst == 0) return HD_OK;
    hipLaunchKernelGGL(gather_cols_kernel, dim3((unsigned)(((size_t)B * Ddst + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, B, Dsrc, Ddst);
    return check("gather_cols");
}

int hd_kvec_matmul(const float* src, const float* mat, float* dst, size_t N, int K, void* stream) {
    if (!src || !mat || !dst || K < 1 || K > 64 || src == dst) return HD_EINVAL;
    if (N == 0) return HD_OK;
    hipLaunchKernelGGL(kvec_matmul_kernel, dim3((unsigned)((N * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, mat, dst, N, K);
    return check("kvec_matmul");
}

int hd_sandwich_matmul(const float* A, const float* x, const float* Bm, float* dst, int n, int S, void* stream) {
    if (!A || !x || !Bm || !dst || n < 0 || S < 1 || S > 64 || x == dst) return HD_EINVAL;
    if (n == 0) return HD_OK;
    if (!hd_raise_dynamic_lds((const void*)sandwich_matmul_kernel, 64 * 1024)) { hd_set_error("sandwich_matmul: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); return HD_EHIP; }
    hipLaunchKernelGGL(sandwich_matmul_kernel, dim3(n), dim3(256), (size_t)4 * S * S * sizeof(float), (hipStream_t)stream, A, x, Bm, dst, S);
    return check("sandwich_matmul");
}

int hd_dense_matmul(const float* src, const float* mat, float* dst, int N, int K, int M, void* stream) {
    if (!src || !mat || !dst || N < 0 || K < 1 || M < 1 || src == dst) return HD_EINVAL;
    if (N == 0) return HD_OK;
    hipLaunchKernelGGL(dense_matmul_kernel, dim3((M + 63) / 64, (N + 63) / 64), dim3(256), 0, (hipStream_t)stream, src, mat, dst, N, K, M);
    return check("dense_matmul");
}

int hd_fwht(float* data, int N, int L, float scale, void* stream) {
    if (!data || N < 0 || L < 2 || L > 4096 || (L & (L - 1))) return HD_EINVAL;
    if (N == 0) return HD_OK;
    hipLaunchKernelGGL(fwht_kernel, dim3(N), dim3(256), (size_t)L * sizeof(float), (hipStream_t)stream, data, L, scale);
    return check("fwht");
}

}  // extern "C"
