// Backward passes of the UNet's attention cores -- components for its training step, each checked against
// torch autograd through a test-only entry of include/hicdiff_hip_debug.h.  The 1x1 projections around them (to_qkv, to_out)
// are ordinary convolutions (data gradient on the forward kernel, weight gradient on the Wgrad component); here are the cores:
//
//   Attention        src/hicdiff.py:229-251   sim = (q scale) k^T, softmax over keys, out = attn v             (mid block, 64 or 25 tokens)
//   LinearAttention  src/hicdiff.py:199-227   q softmax over d, k softmax over tokens, q scale, v / n,
//                                             context = k v^T (d x e per head), out = context^T q
// qkv: NHWC [B][n][3*heads*32] (q | k | v, channel = head*32 + d); dout / dqkv in the same layouts.
#include "hd_common.h"
#include "../../include/hicdiff_hip.h"

namespace {

constexpr int D = 32;
constexpr float SCALE = 0.17677669529663687f;          // dim_head ** -0.5

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string(what) + ": " + hipGetErrorString(e)); return -3; }
    return 0;
}

// One workgroup per (sample, head), n <= 64 tokens: everything lives in LDS, probabilities recomputed.
__global__ __launch_bounds__(256) void attn_full_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, int n, int heads,
                                                            float* __restrict__ dqkv) {
    __shared__ float q[64][D + 1], k[64][D + 1], v[64][D + 1], go[64][D + 1], P[64][65], dS[64][65];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x, C3 = 3 * heads * D, C1 = heads * D;
    const float* base = qkv + (size_t)b * n * C3;
    for (int i = tid; i < n * D; i += 256) {
        const int r = i / D, c = i % D;
        q[r][c] = base[(size_t)r * C3 + h * D + c] * SCALE;
        k[r][c] = base[(size_t)r * C3 + C1 + h * D + c];
        v[r][c] = base[(size_t)r * C3 + 2 * C1 + h * D + c];
        go[r][c] = dout[((size_t)b * n + r) * C1 + h * D + c];
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += 256) {               // scores and dP = dO v^T
        const int i = e / n, j = e % n;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s += q[i][d] * k[j][d]; dp += go[i][d] * v[j][d]; }
        P[i][j] = s; dS[i][j] = dp;
    }
    __syncthreads();
    if (tid < n) {                                           // softmax over keys, then dS = P (dP - sum_j P dP), one query row per thread
        const int i = tid;
        float mx = -3.0e38f;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, P[i][j]);
        float den = 0.f;
        for (int j = 0; j < n; ++j) { const float p = __expf(P[i][j] - mx); P[i][j] = p; den += p; }
        const float inv = 1.f / den;
        float dot = 0.f;
        for (int j = 0; j < n; ++j) { P[i][j] *= inv; dot += P[i][j] * dS[i][j]; }
        for (int j = 0; j < n; ++j) dS[i][j] = P[i][j] * (dS[i][j] - dot);
    }
    __syncthreads();
    float* dbase = dqkv + (size_t)b * n * C3;
    for (int e = tid; e < n * D; e += 256) {
        const int r = e / D, c = e % D;
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < n; ++j) {
            dq += dS[r][j] * k[j][c];                        // row r as query
            dk += dS[j][r] * q[j][c];                        // row r as key (q already carries the scale)
            dv += P[j][r] * go[j][c];
        }
        dbase[(size_t)r * C3 + h * D + c] = dq * SCALE;
        dbase[(size_t)r * C3 + C1 + h * D + c] = dk;
        dbase[(size_t)r * C3 + 2 * C1 + h * D + c] = dv;
    }
}


// ---- LinearAttention ---------------------------------------------------------------------------------------------------------
// Per (sample, head): q' = softmax_d(q) * scale per token; k' = softmax over tokens per d; v' = v / n;
// context[d][e] = sum_n k'[d][n] v'[e][n];  out[e][n] = sum_d context[d][e] q'[d][n].
// Backward in three passes over 64-token chunks (one wave per chunk, one token per lane):
//   1. per chunk: m_c[d], s_c[d] (online-softmax pieces of k), ctx_c[d][e] = sum_n exp(k - m_c) v', dctx_c[d][e] = sum_n q' dout
//   2. per (sample, head): combine -> M, S, context, dcontext, r[d] = sum_e dcontext[d][e] context[d][e]
//      (= sum_n k'[d][n] dk'[d][n]: the softmax-over-tokens correction needs no further sweep)
//   3. per chunk: dq, dk, dv of every token.
// part layout per (sample, head, chunk): [m 32][s 32][ctx 1024][dctx 1024]; fin per (sample, head): [M 32][S 32][r 32][context 1024][dcontext 1024].
constexpr int LA_PART = 64 + 2048, LA_FIN = 96 + 2048, LA_TOK = 64;

__device__ __forceinline__ void la_load_row(const float* p, float (&r)[D]) {
#pragma unroll
    for (int j = 0; j < D; j += 4) { const float4 t = *reinterpret_cast<const float4*>(p + j); r[j] = t.x; r[j + 1] = t.y; r[j + 2] = t.z; r[j + 3] = t.w; }
}
__device__ __forceinline__ void la_softmax_d(float (&q)[D]) {
    float mx = q[0];
#pragma unroll
    for (int j = 1; j < D; ++j) mx = fmaxf(mx, q[j]);
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) { q[j] = __expf(q[j] - mx); den += q[j]; }
    const float inv = 1.f / den;
#pragma unroll
    for (int j = 0; j < D; ++j) q[j] *= inv;
}

__global__ __launch_bounds__(64) void linattn_bwd_partial_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, int n, int heads, int nchunk,
                                                                 float* __restrict__ part) {
    // rows of 36 floats: 16-byte aligned, so a token's row is written and a 4-channel group read as float4 (one b32 access per operand
    // made the pair loop LDS-issue bound: two reads per FMA)
    constexpr int LP = D + 4;
    __shared__ __attribute__((aligned(16))) float kk[LA_TOK][LP], vv[LA_TOK][LP], qq[LA_TOK][LP], dd[LA_TOK][LP];
    __shared__ float mm[D];
    const int ck = blockIdx.x % nchunk, bh = blockIdx.x / nchunk, b = bh / heads, h = bh % heads, lane = threadIdx.x;
    const int C3 = 3 * heads * D, C1 = heads * D, tok = ck * LA_TOK + lane;
    const bool on = tok < n;
    float q[D], k[D], v[D], g[D];
    if (on) {
        const float* row = qkv + ((size_t)b * n + tok) * C3 + h * D;
        la_load_row(row, q); la_load_row(row + C1, k); la_load_row(row + 2 * C1, v);
        la_load_row(dout + ((size_t)b * n + tok) * C1 + h * D, g);
        la_softmax_d(q);
    }
    const float invn = 1.f / (float)n;
#pragma unroll
    for (int j = 0; j < D; j += 4) {
        *reinterpret_cast<float4*>(&kk[lane][j]) = on ? make_float4(k[j], k[j + 1], k[j + 2], k[j + 3]) : make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
        *reinterpret_cast<float4*>(&vv[lane][j]) = on ? make_float4(v[j] * invn, v[j + 1] * invn, v[j + 2] * invn, v[j + 3] * invn) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&qq[lane][j]) = on ? make_float4(q[j] * SCALE, q[j + 1] * SCALE, q[j + 2] * SCALE, q[j + 3] * SCALE) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&dd[lane][j]) = on ? make_float4(g[j], g[j + 1], g[j + 2], g[j + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * LA_PART;
    if (lane < D) {                                          // lane = d: max and sum of exp over the chunk's tokens
        float mx = -3.0e38f;
        for (int t = 0; t < LA_TOK; ++t) mx = fmaxf(mx, kk[t][lane]);
        float s = 0.f;
        for (int t = 0; t < LA_TOK; ++t) { const float e = kk[t][lane] > -1.0e38f ? __expf(kk[t][lane] - mx) : 0.f; kk[t][lane] = e; s += e; }
        mm[lane] = mx; out[lane] = mx; out[D + lane] = s;
    }
    __syncthreads();
    // lane = a 4 x 4 block of (d, e) pairs: four float4 reads per token feed 32 FMAs; each pair still adds its tokens in order
    const int d0 = (lane >> 3) * 4, e0 = (lane & 7) * 4;
    float c[4][4], dc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) { c[a][b2] = 0.f; dc[a][b2] = 0.f; }
    for (int t = 0; t < LA_TOK; ++t) {
        const float4 k4 = *reinterpret_cast<const float4*>(&kk[t][d0]), v4 = *reinterpret_cast<const float4*>(&vv[t][e0]);
        const float4 q4 = *reinterpret_cast<const float4*>(&qq[t][d0]), g4 = *reinterpret_cast<const float4*>(&dd[t][e0]);
        const float ka[4] = {k4.x, k4.y, k4.z, k4.w}, va[4] = {v4.x, v4.y, v4.z, v4.w}, qa[4] = {q4.x, q4.y, q4.z, q4.w}, ga[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) { c[a][b2] += ka[a] * va[b2]; dc[a][b2] += qa[a] * ga[b2]; }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        *reinterpret_cast<float4*>(out + 64 + (d0 + a) * D + e0) = make_float4(c[a][0], c[a][1], c[a][2], c[a][3]);
        *reinterpret_cast<float4*>(out + 64 + 1024 + (d0 + a) * D + e0) = make_float4(dc[a][0], dc[a][1], dc[a][2], dc[a][3]);
    }
}

__global__ __launch_bounds__(256) void linattn_bwd_combine_kernel(const float* __restrict__ part, int nchunk, float* __restrict__ fin) {
    __shared__ float M[D], S[D], red[256];
    const float* p = part + (size_t)blockIdx.x * nchunk * LA_PART;
    float* out = fin + (size_t)blockIdx.x * LA_FIN;
    const int tid = threadIdx.x;
    if (tid < D) {
        float mx = -3.0e38f;
        for (int c = 0; c < nchunk; ++c) mx = fmaxf(mx, p[(size_t)c * LA_PART + tid]);
        float s = 0.f;
        for (int c = 0; c < nchunk; ++c) s += p[(size_t)c * LA_PART + D + tid] * __expf(p[(size_t)c * LA_PART + tid] - mx);
        M[tid] = mx; S[tid] = s; out[tid] = mx; out[D + tid] = s;
    }
    __syncthreads();
    float rpart[4];
    for (int j = 0; j < 4; ++j) {                            // element i = tid + 256 j = (d, e); d = i / 32 is the same for 32 consecutive i
        const int i = tid + 256 * j, d = i / D;
        // four chunks per round with their own accumulators: the loads of a round are independent, so they travel together (one chunk
        // at a time this loop was a chain of 64 dependent round trips on the 64 x 64 maps: 150 us)
        float c4[4] = {0.f, 0.f, 0.f, 0.f}, d4[4] = {0.f, 0.f, 0.f, 0.f};
        int ch = 0;
        for (; ch + 4 <= nchunk; ch += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* q = p + (size_t)(ch + u) * LA_PART;
                c4[u] += q[64 + i] * __expf(q[d] - M[d]);
                d4[u] += q[64 + 1024 + i];
            }
        }
        for (; ch < nchunk; ++ch) {
            const float* q = p + (size_t)ch * LA_PART;
            c4[0] += q[64 + i] * __expf(q[d] - M[d]);
            d4[0] += q[64 + 1024 + i];
        }
        float c = (c4[0] + c4[1]) + (c4[2] + c4[3]);
        const float dc = (d4[0] + d4[1]) + (d4[2] + d4[3]);
        c /= S[d];
        out[96 + i] = c; out[96 + 1024 + i] = dc;
        rpart[j] = c * dc;
    }
    // r[d] = sum_e context[d][e] dcontext[d][e]: 32 consecutive threads of each j hold one d
    for (int j = 0; j < 4; ++j) {
        red[tid] = rpart[j];
        __syncthreads();
        if ((tid & 31) == 0) { float s = 0.f; for (int e = 0; e < D; ++e) s += red[tid + e]; out[2 * D + (tid + 256 * j) / D] = s; }
        __syncthreads();
    }
}

// (Register caps were tried: 168 / 128 registers per wave spill 144 / 304 of them and the kernel goes 127 -> 176 / 305 us; uncapped it uses ~290.)
__global__ __launch_bounds__(64) void linattn_bwd_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ fin, int n,
                                                               int heads, int nchunk, float* __restrict__ dqkv) {
    // rows of 32 floats, 16-byte aligned: every lane reads the same address (a broadcast: no conflicts at any pitch), and the unrolled loops'
    // constant offsets then merge into ds_read_b128
    __shared__ __attribute__((aligned(16))) float ctx[D][D], dctx[D][D];
    __shared__ float M[D], S[D], R[D];
    const int ck = blockIdx.x % nchunk, bh = blockIdx.x / nchunk, b = bh / heads, h = bh % heads, lane = threadIdx.x;
    const float* f = fin + (size_t)bh * LA_FIN;
    for (int i = lane; i < D * D; i += 64) { ctx[i / D][i % D] = f[96 + i]; dctx[i / D][i % D] = f[96 + 1024 + i]; }
    if (lane < D) { M[lane] = f[lane]; S[lane] = f[D + lane]; R[lane] = f[2 * D + lane]; }
    __syncthreads();
    const int C3 = 3 * heads * D, C1 = heads * D, tok = ck * LA_TOK + lane;
    if (tok >= n) return;
    const float* row = qkv + ((size_t)b * n + tok) * C3 + h * D;
    // two phases with disjoint register sets (q, dout, t -> dq; then k, v, dv -> dk, dv)
    const float invn = 1.f / (float)n;
    float* drow = dqkv + ((size_t)b * n + tok) * C3 + h * D;
    {
        float q[D], g[D];
        la_load_row(row, q);
        la_load_row(dout + ((size_t)b * n + tok) * C1 + h * D, g);
        la_softmax_d(q);
        float t[D], dot = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {                        // d q'[d] = sum_e context[d][e] dout[e]; times scale
            __builtin_amdgcn_sched_barrier(0);               // keep each row's LDS reads inside its iteration
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < D; ++e) s += ctx[d][e] * g[e];
            t[d] = s * SCALE; dot += q[d] * t[d];
        }
#pragma unroll
        for (int d = 0; d < D; d += 4)
            *reinterpret_cast<float4*>(drow + d) = make_float4(q[d] * (t[d] - dot), q[d + 1] * (t[d + 1] - dot), q[d + 2] * (t[d + 2] - dot), q[d + 3] * (t[d + 3] - dot));
    }
    __builtin_amdgcn_sched_barrier(0);                       // the second phase's rows are requested after the first phase has retired its registers
    float k[D], v[D];
    la_load_row(row + C1, k); la_load_row(row + 2 * C1, v);
    float kp[D], dv[D];
#pragma unroll
    for (int e = 0; e < D; ++e) dv[e] = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        __builtin_amdgcn_sched_barrier(0);
        kp[d] = __expf(k[d] - M[d]) / S[d];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < D; ++e) { s += dctx[d][e] * v[e]; dv[e] += kp[d] * dctx[d][e]; }
        k[d] = kp[d] * (s * invn - R[d]);                    // d k[d]
    }
#pragma unroll
    for (int d = 0; d < D; d += 4) {
        *reinterpret_cast<float4*>(drow + C1 + d) = make_float4(k[d], k[d + 1], k[d + 2], k[d + 3]);
        *reinterpret_cast<float4*>(drow + 2 * C1 + d) = make_float4(dv[d] * invn, dv[d + 1] * invn, dv[d + 2] * invn, dv[d + 3] * invn);
    }
}

}  // namespace

int launch_attn_full_bwd(const float* qkv, const float* dout, int B, int n, int heads, float* dqkv, hipStream_t st) {
    if (n < 1 || n > 64) { hd_set_error("full-attention backward: at most 64 tokens"); return -1; }
    hipLaunchKernelGGL(attn_full_bwd_kernel, dim3(B * heads), dim3(256), 0, st, qkv, dout, n, heads, dqkv);
    return check_launch("attn_full_bwd");
}

size_t linattn_bwd_scratch_floats(int B, int n, int heads) { return (size_t)B * heads * (((size_t)n + LA_TOK - 1) / LA_TOK * LA_PART + LA_FIN); }

int launch_linattn_bwd(const float* qkv, const float* dout, int B, int n, int heads, float* scratch, float* dqkv, hipStream_t st) {
    const int nchunk = (n + LA_TOK - 1) / LA_TOK;
    float* part = scratch;
    float* fin = scratch + (size_t)B * heads * nchunk * LA_PART;
    hipLaunchKernelGGL(linattn_bwd_partial_kernel, dim3(B * heads * nchunk), dim3(64), 0, st, qkv, dout, n, heads, nchunk, part);
    hipLaunchKernelGGL(linattn_bwd_combine_kernel, dim3(B * heads), dim3(256), 0, st, part, nchunk, fin);
    hipLaunchKernelGGL(linattn_bwd_apply_kernel, dim3(B * heads * nchunk), dim3(64), 0, st, qkv, dout, fin, n, heads, nchunk, dqkv);
    return check_launch("linattn_bwd");
}

extern "C" int hd_debug_linattn_bwd(const float* qkv, const float* dout, int B, int n, int heads, float* dqkv, void* stream) {
    if (!qkv || !dout || !dqkv || B < 1 || heads < 1 || n < 1) return HD_EINVAL;
    float* scratch = nullptr;
    if (hipMalloc(&scratch, linattn_bwd_scratch_floats(B, n, heads) * sizeof(float)) != hipSuccess) return HD_ENOMEM;
    const int rc = launch_linattn_bwd(qkv, dout, B, n, heads, scratch, dqkv, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(scratch);
    return rc ? HD_EHIP : HD_OK;
}

extern "C" int hd_debug_attn_full_bwd(const float* qkv, const float* dout, int B, int n, int heads, float* dqkv, void* stream) {
    if (!qkv || !dout || !dqkv || B < 1 || heads < 1) return HD_EINVAL;
    const int rc = launch_attn_full_bwd(qkv, dout, B, n, heads, dqkv, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    return rc ? (rc == -1 ? HD_EINVAL : HD_EHIP) : HD_OK;
}
