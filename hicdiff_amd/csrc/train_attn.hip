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

}  // namespace

int launch_attn_full_bwd(const float* qkv, const float* dout, int B, int n, int heads, float* dqkv, hipStream_t st) {
    if (n < 1 || n > 64) { hd_set_error("full-attention backward: at most 64 tokens"); return -1; }
    hipLaunchKernelGGL(attn_full_bwd_kernel, dim3(B * heads), dim3(256), 0, st, qkv, dout, n, heads, dqkv);
    return check_launch("attn_full_bwd");
}

extern "C" int hd_debug_attn_full_bwd(const float* qkv, const float* dout, int B, int n, int heads, float* dqkv, void* stream) {
    if (!qkv || !dout || !dqkv || B < 1 || heads < 1) return HD_EINVAL;
    const int rc = launch_attn_full_bwd(qkv, dout, B, n, heads, dqkv, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    return rc ? (rc == -1 ? HD_EINVAL : HD_EHIP) : HD_OK;
}
