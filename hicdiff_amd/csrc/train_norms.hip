// Backward passes of the UNet's normalisations -- components for its training step, each checked
// against torch autograd through a test-only entry of include/hicdiff_hip_debug.h.  All reductions run in a fixed order.
//
//   GroupNorm(groups) -> [x (scale+1) + shift] -> SiLU      src/hicdiff.py:155-171 (Block.forward)
//   channel LayerNorm (gain only, biased variance)           src/hicdiff.py:99-108
//   weight standardisation of a conv filter                  src/hicdiff.py:84-97
#include "hd_common.h"
#include "../../include/hicdiff_hip.h"

#include <algorithm>

namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float v) { const float s = sigmoid_f(v); return s * (1.f + v * (1.f - s)); }

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string(what) + ": " + hipGetErrorString(e)); return -3; }
    return 0;
}

// ---- GroupNorm + FiLM + SiLU -------------------------------------------------------------------------------------------
// x: raw conv output [B][HW][C]; g: dL/d(silu output); A, Bv: the forward's per-(sample, channel) affine (v = x A + Bv);
// stats: [B][G][2] (mean, rstd).  part[(b*nchunk + ck)][2][C]: T1 = sum dv, T2 = sum dv * xhat over the chunk's pixels.
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ A,
                                                             const float* __restrict__ Bv, const float* __restrict__ stats, int HW, int C, int G,
                                                             int chunk, float* __restrict__ part) {
    // thread = (channel quad, pixel group): float4 accesses and all 256 threads busy at every width (one thread per channel left three
    // quarters of the workgroup idle at C = 64); the pixel groups' sums meet in LDS and are added in order
    __shared__ float red[2 * 1024];                         // [pixel group][2][C], npg * C = 1024
    const int b = blockIdx.y, ck = blockIdx.x, nchunk = gridDim.x, cg = C / G;
    const int p0 = ck * chunk, p1 = min(HW, p0 + chunk);
    const int nq = C >> 2;
    if (nq <= 256 && (C & 3) == 0 && (cg & 3) == 0) {
        const int npg = 256 / nq, q = threadIdx.x % nq, pg = threadIdx.x / nq, c = q * 4;
        if (pg < npg) {
            const float4 a = *reinterpret_cast<const float4*>(A + (size_t)b * C + c), bb = *reinterpret_cast<const float4*>(Bv + (size_t)b * C + c);
            const float mu = stats[((size_t)b * G + c / cg) * 2], rs = stats[((size_t)b * G + c / cg) * 2 + 1];     // cg % 4 == 0: one group per quad
            float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
            for (int p = p0 + pg; p < p1; p += npg) {
                const size_t e = ((size_t)b * HW + p) * C + c;
                const float4 xv = *reinterpret_cast<const float4*>(x + e), gv = *reinterpret_cast<const float4*>(g + e);
                float dv;
                dv = gv.x * dsilu_f(xv.x * a.x + bb.x); t1.x += dv; t2.x += dv * (xv.x - mu) * rs;
                dv = gv.y * dsilu_f(xv.y * a.y + bb.y); t1.y += dv; t2.y += dv * (xv.y - mu) * rs;
                dv = gv.z * dsilu_f(xv.z * a.z + bb.z); t1.z += dv; t2.z += dv * (xv.z - mu) * rs;
                dv = gv.w * dsilu_f(xv.w * a.w + bb.w); t1.w += dv; t2.w += dv * (xv.w - mu) * rs;
            }
            *reinterpret_cast<float4*>(red + (size_t)pg * 2 * C + c) = t1;
            *reinterpret_cast<float4*>(red + (size_t)pg * 2 * C + C + c) = t2;
        }
        __syncthreads();
        float* d = part + (size_t)(b * nchunk + ck) * 2 * C;
        for (int i = threadIdx.x; i < 2 * C; i += 256) {
            float t = 0.f;
            for (int k = 0; k < npg; ++k) t += red[(size_t)k * 2 * C + i];
            d[i] = t;
        }
        return;
    }
    for (int c = threadIdx.x; c < C; c += 256) {            // wider than 1024 channels: one thread per channel
        const float a = A[(size_t)b * C + c], bb = Bv[(size_t)b * C + c];
        const float mu = stats[((size_t)b * G + c / cg) * 2], rs = stats[((size_t)b * G + c / cg) * 2 + 1];
        float t1 = 0.f, t2 = 0.f;
        for (int p = p0; p < p1; ++p) {
            const size_t e = ((size_t)b * HW + p) * C + c;
            const float xv = x[e];
            const float dv = g[e] * dsilu_f(xv * a + bb);
            t1 += dv; t2 += dv * (xv - mu) * rs;
        }
        float* d = part + (size_t)(b * nchunk + ck) * 2 * C;
        d[c] = t1; d[C + c] = t2;
    }
}

// One workgroup per sample.  part: [B][nchunk][2][C] per-chunk sums, added here in col_sum_kernel's order (eight chains).  film: [B][film_bs] with scale at film_off + c and shift at
// film_off + C + c (film_mode 1), or nothing (0).  Writes coef[b][c] = {rstd w, rstd M1_g, rstd M2_g, -} for the apply pass,
// dfilm[b][2C] = (d scale, d shift) and U[b][2][C] = ((1+s) T2, (1+s) T1) whose sums over b are d gamma, d beta.
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ part, int nchunk, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ film,
                                                              int film_bs, int film_off, int film_mode, int HW, int C, int G, float4* __restrict__ coef,
                                                              float* __restrict__ dfilm, float* __restrict__ U, int dfilm_bs) {
    extern __shared__ float Tsh[];                           // [2][C]: sum dv, sum dv * xhat of this sample
    __shared__ float m1[64], m2[64];
    const int b = blockIdx.x, cg = C / G;
    for (int col = threadIdx.x; col < 2 * C; col += 256) {
        const float* in = part + (size_t)b * nchunk * 2 * C + col;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int r = 0;
        for (; r + 8 <= nchunk; r += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += in[(size_t)(r + j) * 2 * C];
        }
        for (; r < nchunk; ++r) a[r & 7] += in[(size_t)r * 2 * C];
        Tsh[col] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    __syncthreads();
    if (threadIdx.x < G) {                                   // one thread per group walks its channels in order
        const int g0 = threadIdx.x * cg;
        float s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < cg; ++j) {
            const int c = g0 + j;
            const float w = gamma[c] * (film_mode == 1 ? film[(size_t)b * film_bs + film_off + c] + 1.f : 1.f);
            s1 += w * Tsh[c]; s2 += w * Tsh[C + c];
        }
        const float n = (float)HW * cg;
        m1[threadIdx.x] = s1 / n; m2[threadIdx.x] = s2 / n;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int gi = c / cg;
        const float rs = stats[((size_t)b * G + gi) * 2 + 1];
        const float sc1 = film_mode == 1 ? film[(size_t)b * film_bs + film_off + c] + 1.f : 1.f;
        const float t1 = Tsh[c], t2 = Tsh[C + c];
        coef[(size_t)b * C + c] = make_float4(rs * gamma[c] * sc1, rs * m1[gi], rs * m2[gi], 0.f);
        if (dfilm && film_mode == 1) { dfilm[(size_t)b * dfilm_bs + c] = gamma[c] * t2 + beta[c] * t1; dfilm[(size_t)b * dfilm_bs + C + c] = t1; }
        U[((size_t)b * 2) * C + c] = sc1 * t2; U[((size_t)b * 2 + 1) * C + c] = sc1 * t1;
    }
}

// gout <- dL/dx = rstd (dv w - M1 - xhat M2); gout may be g
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* g, float* gout, const float* __restrict__ A,
                                                           const float* __restrict__ Bv, const float* __restrict__ stats, const float4* __restrict__ coef,
                                                           int HW, int C, int G, size_t n) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int c = (int)(e % C), b = (int)(e / ((size_t)HW * C)), cg = C / G;
    const float xv = x[e];
    const float dv = g[e] * dsilu_f(xv * A[(size_t)b * C + c] + Bv[(size_t)b * C + c]);
    const float mu = stats[((size_t)b * G + c / cg) * 2], rs = stats[((size_t)b * G + c / cg) * 2 + 1];
    const float4 k = coef[(size_t)b * C + c];
    gout[e] = dv * k.x - k.y - (xv - mu) * rs * k.z;
}

// out[col] (+)= sum_row in[row][col], eight independent chains; grid.y batches (in and out advance by nrows*ncols and ncols)
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ in, int nrows, int ncols, int accumulate, float* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols) return;
    in += (size_t)blockIdx.y * nrows * ncols; out += (size_t)blockIdx.y * ncols;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = 0;
    for (; r + 8 <= nrows; r += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += in[(size_t)(r + j) * ncols + col];
    }
    for (; r < nrows; ++r) a[r & 7] += in[(size_t)r * ncols + col];
    const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    out[col] = (accumulate ? out[col] : 0.f) + s;
}

// out[g][col] = sum of rows g*per .. min(nrows, (g+1)*per) - 1 of in[row][col]  (first level of a long column sum)
__global__ __launch_bounds__(256) void col_sum_groups_kernel(const float* __restrict__ in, int nrows, int per, int ncols, float* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (col >= ncols) return;
    const int r0 = g * per, r1 = min(nrows, r0 + per);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += in[(size_t)(r + j) * ncols + col];
    }
    for (; r < r1; ++r) a[0] += in[(size_t)r * ncols + col];
    out[(size_t)g * ncols + col] = (a[0] + a[1]) + (a[2] + a[3]);
}

// dgamma[c] (+)= sum_b U[b][0][c], dbeta[c] (+)= sum_b U[b][1][c]  (col_sum_kernel's order over b)
__global__ __launch_bounds__(256) void gn_param_grads_kernel(const float* __restrict__ U, int Bn, int C, int accumulate, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= 2 * C) return;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = 0;
    for (; r + 8 <= Bn; r += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += U[(size_t)(r + j) * 2 * C + col];
    }
    for (; r < Bn; ++r) a[r & 7] += U[(size_t)r * 2 * C + col];
    const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    float* out = col < C ? dgamma + col : dbeta + (col - C);
    *out = (accumulate ? *out : 0.f) + s;
}

// ---- channel LayerNorm (gain only) ---------------------------------------------------------------------------------------
// y = (x - mean_c) rstd g;  dx = rstd (dy g - mean_c(dy g) - xhat mean_c(dy g xhat)) written to dout (may be dy).
// A row's C channels are spread over LPR = min(64, C / 4) lanes, a float4 (two when C = 512) per lane: a wave works on 64 / LPR rows at
// once (four at C = 64; round 1's one-row-per-wave, one-channel-per-lane form ran at a third of the memory rate there).  Each lane
// keeps its channels' d gain sums in registers; lanes, waves and then workgroups (part[blockIdx][C], col_sum afterwards) are combined
// in a fixed order.  NV = float4s per lane.
// add (optional, may be dout): dout = add + dx -- the gradient accumulates into a tensor that already holds other contributions.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* dy, float* dout, const float* __restrict__ gain, size_t P, int C,
                                                     int rows_per_block, float* __restrict__ part, const float* add) {
    extern __shared__ float dg[];                           // [4 waves][64 / LPR rows][C]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lpr = C / (4 * NV), rpw = 64 / lpr;          // lanes per row, rows per wave
    const int sub = lane % lpr, rw = lane / lpr;
    float4 gq[NV], acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { gq[v] = *reinterpret_cast<const float4*>(gain + (sub + v * lpr) * 4); acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); }
    const size_t r0 = (size_t)blockIdx.x * rows_per_block;
    const float invC = 1.f / C;
    for (int rr = w * rpw + rw; rr < rows_per_block; rr += 4 * rpw) {
        const size_t r = r0 + rr;
        const bool live = r < P;                              // whole-wave shuffles below: dead rows compute on zeros and store nothing
        float4 xv[NV], dv[NV];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            xv[v] = live ? *reinterpret_cast<const float4*>(x + r * C + (sub + v * lpr) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            dv[v] = live ? *reinterpret_cast<const float4*>(dy + r * C + (sub + v * lpr) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (xv[v].x + xv[v].y) + (xv[v].z + xv[v].w);
        }
        for (int m = lpr >> 1; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        const float mean = s * invC;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            xv[v].x -= mean; xv[v].y -= mean; xv[v].z -= mean; xv[v].w -= mean;
            q += (xv[v].x * xv[v].x + xv[v].y * xv[v].y) + (xv[v].z * xv[v].z + xv[v].w * xv[v].w);
        }
        for (int m = lpr >> 1; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
        const float rs = 1.f / sqrtf(q * invC + 1e-5f);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            xv[v].x *= rs; xv[v].y *= rs; xv[v].z *= rs; xv[v].w *= rs;                 // xhat
            acc[v].x += dv[v].x * xv[v].x; acc[v].y += dv[v].y * xv[v].y; acc[v].z += dv[v].z * xv[v].z; acc[v].w += dv[v].w * xv[v].w;
            dv[v].x *= gq[v].x; dv[v].y *= gq[v].y; dv[v].z *= gq[v].z; dv[v].w *= gq[v].w;   // dy g
            s1 += (dv[v].x + dv[v].y) + (dv[v].z + dv[v].w);
            s2 += (dv[v].x * xv[v].x + dv[v].y * xv[v].y) + (dv[v].z * xv[v].z + dv[v].w * xv[v].w);
        }
        for (int m = lpr >> 1; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
        s1 *= invC; s2 *= invC;
        if (live) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float4 o = make_float4(rs * (dv[v].x - s1 - xv[v].x * s2), rs * (dv[v].y - s1 - xv[v].y * s2), rs * (dv[v].z - s1 - xv[v].z * s2),
                                       rs * (dv[v].w - s1 - xv[v].w * s2));
                if (add) { const float4 t = *reinterpret_cast<const float4*>(add + r * C + (sub + v * lpr) * 4); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
                *reinterpret_cast<float4*>(dout + r * C + (sub + v * lpr) * 4) = o;
            }
        }
    }
    // d gain: [wave][row slot][C] in LDS, then one thread per channel adds the 4 * rpw slots in order
#pragma unroll
    for (int v = 0; v < NV; ++v) *reinterpret_cast<float4*>(dg + ((size_t)(w * rpw + rw) * C) + (sub + v * lpr) * 4) = acc[v];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = 0.f;
        for (int k = 0; k < 4 * rpw; ++k) t += dg[(size_t)k * C + c];
        part[(size_t)blockIdx.x * C + c] = t;
    }
}

// ---- weight standardisation ------------------------------------------------------------------------------------------------
// What = (W - mean) rsqrt(var + 1e-5) per output filter (biased variance over its n = Cin*KH*KW entries);
// dW = rstd (dWhat - mean(dWhat) - What mean(dWhat What)).  One workgroup per filter; fp64 sums as in pack_conv_kernel.
__global__ __launch_bounds__(256) void ws_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dwhat, int n, float* __restrict__ dw) {
    __shared__ double red[256];
    const float* s = w + (size_t)blockIdx.x * n;
    const float* d = dwhat + (size_t)blockIdx.x * n;
    auto total = [&](double v) {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int m = 128; m > 0; m >>= 1) { if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m]; __syncthreads(); }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += s[i];
    const double mean = total(acc) / n;
    acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const double t = s[i] - mean; acc += t * t; }
    const double rstd = 1.0 / sqrt(total(acc) / n + 1e-5);
    double a1 = 0.0, a2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const double wh = (s[i] - mean) * rstd; a1 += d[i]; a2 += d[i] * wh; }
    const double m1 = total(a1) / n, m2 = total(a2) / n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double wh = (s[i] - mean) * rstd;
        dw[(size_t)blockIdx.x * n + i] = (float)(rstd * (d[i] - m1 - wh * m2));
    }
}

// What (torch layout) of a filter bank, the same arithmetic as pack_conv_kernel's standardize
__global__ __launch_bounds__(256) void ws_fwd_kernel(const float* __restrict__ w, int n, float* __restrict__ out) {
    __shared__ double red[256];
    const float* s = w + (size_t)blockIdx.x * n;
    auto total = [&](double v) {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int m = 128; m > 0; m >>= 1) { if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m]; __syncthreads(); }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += s[i];
    const double mean = total(acc) / n;
    acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const double t = s[i] - mean; acc += t * t; }
    const double rstd = 1.0 / sqrt(total(acc) / n + 1e-5);
    for (int i = threadIdx.x; i < n; i += 256) out[(size_t)blockIdx.x * n + i] = (float)((s[i] - mean) * rstd);
}

}  // namespace

int launch_ws_fwd(const float* w, int Cout, int n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(ws_fwd_kernel, dim3(Cout), dim3(256), 0, st, w, n, out);
    return check_launch("ws forward");
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
// scratch: B * nchunk * 2C (partials) + B * 2C (T) + B * 2C (U) + B * C * 4 (coef) floats, nchunk = ceil(HW / 64)
size_t gn_bwd_scratch_floats(int B, int HW, int C) { return (size_t)B * ((HW + 63) / 64) * 2 * C + (size_t)B * 2 * C * 2 + (size_t)B * C * 4; }

int launch_gn_silu_bwd(const float* x, const float* g, float* gout, const float* A, const float* Bv, const float* stats, const float* gamma, const float* beta,
                       const float* film, int film_bs, int film_off, int film_mode, int B, int HW, int C, int G, float* scratch, float* dgamma, float* dbeta,
                       float* dfilm, int accumulate, hipStream_t st, int dfilm_bs) {
    if (dfilm_bs <= 0) dfilm_bs = 2 * C;                       // dfilm rows: (d scale | d shift) at dfilm + b * dfilm_bs
    if (C % G || G > 64 || (film_mode != 0 && film_mode != 1)) { hd_set_error("gn backward: unsupported shape"); return -1; }
    const int nchunk = (HW + 63) / 64;
    float* part = scratch;
    float* U = part + (size_t)B * nchunk * 2 * C + (size_t)B * 2 * C;
    float4* coef = reinterpret_cast<float4*>(U + (size_t)B * 2 * C);
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nchunk, B), dim3(256), 0, st, x, g, A, Bv, stats, HW, C, G, 64, part);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(256), (size_t)2 * C * sizeof(float), st, part, nchunk, stats, gamma, beta, film, film_bs, film_off,
                       film_mode, HW, C, G, coef, dfilm, U, dfilm_bs);
    const size_t n = (size_t)B * HW * C;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, g, gout, A, Bv, stats, coef, HW, C, G, n);
    hipLaunchKernelGGL(gn_param_grads_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, U, B, C, accumulate, dgamma, dbeta);
    return check_launch("gn backward");
}

// scratch: one d-gain row per workgroup of 256 pixel rows, and 32 rows for the first level of their sum
size_t ln_bwd_scratch_floats(size_t P, int C) { return ((P + 255) / 256 + 32) * (size_t)C; }

int launch_ln_bwd(const float* x, const float* dy, float* dout, const float* gain, size_t P, int C, float* scratch, float* dgain, int accumulate, hipStream_t st,
                  const float* add) {
    if (C % 64 || C > 512 || (C > 256 && C != 512)) { hd_set_error("ln backward: channels must be 64, 128, 192, 256 or 512"); return -1; }
    const int rows = 256;
    const unsigned nb = (unsigned)((P + rows - 1) / rows);
    const int nv = C == 512 ? 2 : 1, lpr = C / (4 * nv);
    if (64 % lpr) { hd_set_error("ln backward: unsupported channel count"); return -1; }
    const size_t lds = (size_t)4 * (64 / lpr) * C * sizeof(float);
    if (nv == 2) hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(nb), dim3(256), lds, st, x, dy, dout, gain, P, C, rows, scratch, add);
    else hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(nb), dim3(256), lds, st, x, dy, dout, gain, P, C, rows, scratch, add);
    // sum over the workgroups' rows: two levels when there are many (one thread per column walking thousands of rows was 47 us)
    if (nb >= 256) {
        float* lvl = scratch + (size_t)nb * C;
        const int per = (int)((nb + 31) / 32);
        hipLaunchKernelGGL(col_sum_groups_kernel, dim3((C + 255) / 256, 32), dim3(256), 0, st, scratch, (int)nb, per, C, lvl);
        hipLaunchKernelGGL(col_sum_kernel, dim3((C + 255) / 256), dim3(256), 0, st, lvl, 32, C, accumulate, dgain);
    } else {
        hipLaunchKernelGGL(col_sum_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, (int)nb, C, accumulate, dgain);
    }
    return check_launch("ln backward");
}

int launch_ws_bwd(const float* w, const float* dwhat, int Cout, int n, float* dw, hipStream_t st) {
    hipLaunchKernelGGL(ws_bwd_kernel, dim3(Cout), dim3(256), 0, st, w, dwhat, n, dw);
    return check_launch("ws backward");
}

// ---- test-only entries (include/hicdiff_hip_debug.h) ------------------------------------------------------------------------
extern "C" {

int hd_debug_gn_silu_bwd(const float* x, float* g, const float* gamma, const float* beta, const float* film, int B, int H, int W, int C, int G,
                         float* dgamma, float* dbeta, float* dfilm, void* stream) {
    if (!x || !g || !gamma || !beta || !dgamma || !dbeta || B < 1 || C % G) return HD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W;
    int slots = 0;
    float *part = nullptr, *A = nullptr, *Bv = nullptr, *stats = nullptr, *scratch = nullptr;
    const size_t pslots = (size_t)(HW + 255) / 256;
    bool ok = hipMalloc(&part, (size_t)B * pslots * C * 2 * sizeof(float)) == hipSuccess && hipMalloc(&A, (size_t)B * C * sizeof(float)) == hipSuccess &&
              hipMalloc(&Bv, (size_t)B * C * sizeof(float)) == hipSuccess && hipMalloc(&stats, (size_t)B * G * 2 * sizeof(float)) == hipSuccess &&
              hipMalloc(&scratch, gn_bwd_scratch_floats(B, HW, C) * sizeof(float)) == hipSuccess;
    int rc = ok ? 0 : -4;
    if (!rc) rc = launch_gn_partial(x, B, HW, C, part, &slots, st);
    if (!rc) rc = launch_gn_finalize(part, slots, B, HW, C, G, gamma, beta, film, 2 * C, 0, film ? 1 : 0, A, Bv, nullptr, st, stats);
    if (!rc) rc = launch_gn_silu_bwd(x, g, g, A, Bv, stats, gamma, beta, film, 2 * C, 0, film ? 1 : 0, B, HW, C, G, scratch, dgamma, dbeta, dfilm, 0, st, 0);
    (void)hipStreamSynchronize(st);
    for (float* p : {part, A, Bv, stats, scratch}) if (p) (void)hipFree(p);
    return rc ? (rc == -4 ? HD_ENOMEM : HD_EHIP) : HD_OK;
}

int hd_debug_ln_bwd(const float* x, float* dy, const float* gain, long long P, int C, float* dgain, void* stream) {
    if (!x || !dy || !gain || !dgain || P < 1 || C < 1) return HD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float* scratch = nullptr;
    if (hipMalloc(&scratch, ln_bwd_scratch_floats((size_t)P, C) * sizeof(float)) != hipSuccess) return HD_ENOMEM;
    const int rc = launch_ln_bwd(x, dy, dy, gain, (size_t)P, C, scratch, dgain, 0, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(scratch);
    return rc ? HD_EHIP : HD_OK;
}

int hd_debug_ws_bwd(const float* w, const float* dwhat, int Cout, int n, float* dw, void* stream) {
    if (!w || !dwhat || !dw || Cout < 1 || n < 1) return HD_EINVAL;
    const int rc = launch_ws_bwd(w, dwhat, Cout, n, dw, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    return rc ? HD_EHIP : HD_OK;
}

}  // extern "C"
