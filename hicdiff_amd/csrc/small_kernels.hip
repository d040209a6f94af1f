// HBM-bound and tiny kernels of the HiCDiff engine (gfx950): first/last convolutions with one or two
// input channels, time embedding + FiLM projections, GroupNorm statistics, channel LayerNorm, linear
// and full attention, and the fused sampler updates.  Activations are NHWC fp32.
#include "hd_common.h"
#include <cstdlib>

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

static inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string(what) + ": " + hipGetErrorString(e)); return -3; }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// First convolution: Cin = 1 (x) or 2 (cat(cond, x), src/hicdiff.py:352), KS = 7 (Unet.init_conv
// src/hicdiff.py:279) or 3 (hicedrn head src/model/hicedrn_Diff.py:226).  Pure bandwidth: one
// thread per output pixel, weights broadcast from LDS, NHWC float4 stores.
// w: torch layout [Cout][Cin][KS][KS].
template <int KS>
__global__ __launch_bounds__(256) void conv_small_cin_kernel(const float* __restrict__ x, const float* __restrict__ cond,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ out, int B, int S, int Cin, int Cout) {
    constexpr int T = 16, L = T + KS - 1, TAPS = KS * KS, R = KS / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);            // [Cin*TAPS][Cout]
    float* tile = wl + Cin * TAPS * Cout;                  // [Cin][L][L]
    const int tid = threadIdx.x;
    const int tiles = (S + T - 1) / T;
    const int b = blockIdx.x / (tiles * tiles);
    const int tr = blockIdx.x % (tiles * tiles);
    const int y0 = (tr / tiles) * T, x0 = (tr % tiles) * T;
    for (int i = tid; i < Cin * TAPS * Cout; i += 256) {
        int co = i % Cout, kt = i / Cout;                  // kt = ci*TAPS + tap
        wl[i] = w[(size_t)co * Cin * TAPS + kt];
    }
    for (int i = tid; i < Cin * L * L; i += 256) {
        int ci = i / (L * L), r = i % (L * L);
        int yy = y0 + r / L - R, xx = x0 + r % L - R;
        const float* src = (Cin == 2 && ci == 0) ? cond : x;
        tile[i] = (yy >= 0 && yy < S && xx >= 0 && xx < S) ? src[((size_t)b * S + yy) * S + xx] : 0.f;
    }
    __syncthreads();
    const int ty = tid / T, tx = tid % T;
    const int y = y0 + ty, xo = x0 + tx;
    if (y >= S || xo >= S) return;
    float* o = out + (((size_t)b * S + y) * S + xo) * Cout;
    for (int co0 = 0; co0 < Cout; co0 += 16) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias[co0 + j];
        for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const float xv = tile[(ci * L + ty + tap / KS) * L + tx + tap % KS];
                const float4* wp = reinterpret_cast<const float4*>(wl + (ci * TAPS + tap) * Cout + co0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 wv = wp[j];
                    acc[4 * j + 0] += xv * wv.x; acc[4 * j + 1] += xv * wv.y;
                    acc[4 * j + 2] += xv * wv.z; acc[4 * j + 3] += xv * wv.w;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            reinterpret_cast<float4*>(o + co0)[j] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
    }
}

// The same convolution with the LANE on the output channel (round 4; Cout a multiple of 64).  The pixel-per-thread form above is bound by its
// LDS weight reads (one ds_read_b128 per four FMAs) and writes each 256-byte output row as sixteen 16-byte pieces from sixteen instructions:
// 145 us for the 268 MB of the UNet's init_conv at 256 x 64 x 64, 0.23 of the HBM roofline.  Here a wave keeps its 64 channels' filter in
// registers (two taps of one filter row per register pair), walks 8-pixel row segments of an 8 x 32 pixel tile whose zero-padded window sits
// in LDS twice (as it is, and shifted left by one float, so that every pair of neighbouring pixels is an aligned 8-byte read), and issues
// v_pk_fma_f32: acc.lo += w[dx] x[p + dx], acc.hi += w[dx + 1] x[p + dx + 1] -- 28 packed FMAs per pixel and 64 channels instead of 3136 / 64 * 64
// scalar ones, every LDS read a broadcast of 16 window values per 32 FMAs.  A pixel's 64 channels leave as one 256-byte store.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KS, int CIN>
__global__ __launch_bounds__(256) void conv_first_lanes_kernel(const float* __restrict__ x, const float* __restrict__ cond,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ out, int B, int S, int Cout) {
    constexpr int R = KS / 2, TH = 8, TW = 32, LH = TH + KS - 1, LW = 40, NP = (KS + 1) / 2;
    static_assert(TW + KS - 1 <= LW - 2 && KS <= 7, "a 16-float read at column 24 must stay inside the row");
    __shared__ __attribute__((aligned(16))) float win[CIN][2][LH][LW];        // [.][1]: the same rows shifted left by one float
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (S + TW - 1) / TW, tiles_y = (S + TH - 1) / TH;
    const int b = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x % (tiles_x * tiles_y);
    const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) * TW;
    const int co = blockIdx.y * 64 + lane;

    // the window: every thread's elements are requested before the first one is written to LDS (a constant trip count and clamped addresses --
    // as a `for (i = tid; i < N; i += 256)` loop with the load under the bounds test, hipcc kept it rolled: one exposed round trip per element)
    {
        constexpr int NW = CIN * 2 * LH * LW, NIT = (NW + 255) / 256;
        float wv[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = min(tid + k * 256, NW - 1);
            const int c = i % LW, r = (i / LW) % LH, sh = (i / (LW * LH)) & 1, ci = i / (2 * LW * LH);
            const int yy = y0 + r - R, xx = x0 + c + sh - R;
            const float* src = (CIN == 2 && ci == 0) ? cond : x;
            const bool in = yy >= 0 && yy < S && xx >= 0 && xx < S;
            const float v = src[((size_t)b * S + (in ? yy : 0)) * S + (in ? xx : 0)];
            wv[k] = in ? v : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (tid + k * 256 < NW) (&win[0][0][0][0])[tid + k * 256] = wv[k];
    }
    // this lane's filter, torch layout [Cout][CIN][KS][KS]: pairs (w[dy][2k], w[dy][2k + 1]), the last pair of an odd row padded with 0
    f32x2 wp[CIN][KS][NP];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int dy = 0; dy < KS; ++dy)
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const float* wr = w + ((size_t)(co * CIN + ci) * KS + dy) * KS;
                wp[ci][dy][k] = f32x2{wr[2 * k], 2 * k + 1 < KS ? wr[2 * k + 1] : 0.f};
            }
    const float bv = bias[co];
    __syncthreads();

    for (int task = wave; task < TH * (TW / 8); task += 4) {          // 8-pixel row segments; the segment of a wave is uniform
        const int r = task >> 2, cs = task & 3, y = y0 + r;
        if (y >= S || x0 + cs * 8 >= S) continue;
        f32x2 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = f32x2{bv, 0.f};
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int dy = 0; dy < KS; ++dy) {
                float xe[16], xo[16];                                   // xe[m] = window[m], xo[m] = window[m + 1] (columns from cs * 8)
                const float4* pe = reinterpret_cast<const float4*>(&win[ci][0][r + dy][cs * 8]);
                const float4* po = reinterpret_cast<const float4*>(&win[ci][1][r + dy][cs * 8]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = pe[q], c = po[q];
                    xe[4 * q] = a.x; xe[4 * q + 1] = a.y; xe[4 * q + 2] = a.z; xe[4 * q + 3] = a.w;
                    xo[4 * q] = c.x; xo[4 * q + 1] = c.y; xo[4 * q + 2] = c.z; xo[4 * q + 3] = c.w;
                }
#pragma unroll
                for (int k = 0; k < NP; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int m = j + 2 * k;                        // window column of the pair's first tap; m + 1 <= 14
                        const f32x2 xv = (m & 1) ? f32x2{xo[m - 1], xo[m]} : f32x2{xe[m], xe[m + 1]};
                        acc[j] = __builtin_elementwise_fma(wp[ci][dy][k], xv, acc[j]);
                    }
                __builtin_amdgcn_sched_barrier(0);      // one filter row at a time: left alone, hipcc hoists every window read of the task (256 registers)
            }
        float* o = out + (((size_t)b * S + y) * S + x0 + cs * 8) * Cout + co;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[(size_t)j * Cout] = acc[j].x + acc[j].y;     // S % 8 == 0 (launcher): a segment is inside the row as a whole
                                                                                 // (a per-pixel guard makes hipcc sink each pixel's FMAs into its branch: 254 registers)
    }
}

// lanes_ok: the caller's arithmetic mode allows the lane-per-channel kernel (the exact-fp32 mode keeps the summation order its parity
// margins were measured with -- DDIM amplifies a reordering of 1e-7 to 1e-3)
int launch_conv_small_cin(const float* x, const float* cond, const float* w, const float* bias, float* out, int B, int S,
                          int KS, int Cin, int Cout, hipStream_t st, bool lanes_ok) {
    if ((KS != 3 && KS != 7) || Cout % 16 != 0 || Cin < 1 || Cin > 2) { hd_set_error("conv_small_cin: unsupported shape"); return -1; }
    static const bool lanes_off = getenv("HICDIFF_FIRST_OLD") && atoi(getenv("HICDIFF_FIRST_OLD")) != 0;      // A/B switch
    if (lanes_ok && !lanes_off && Cout % 64 == 0 && S % 8 == 0) {
        const dim3 grid((unsigned)(B * ((S + 31) / 32) * ((S + 7) / 8)), (unsigned)(Cout / 64));
        if (KS == 7 && Cin == 1) hipLaunchKernelGGL((conv_first_lanes_kernel<7, 1>), grid, dim3(256), 0, st, x, cond, w, bias, out, B, S, Cout);
        else if (KS == 7) hipLaunchKernelGGL((conv_first_lanes_kernel<7, 2>), grid, dim3(256), 0, st, x, cond, w, bias, out, B, S, Cout);
        else if (Cin == 1) hipLaunchKernelGGL((conv_first_lanes_kernel<3, 1>), grid, dim3(256), 0, st, x, cond, w, bias, out, B, S, Cout);
        else hipLaunchKernelGGL((conv_first_lanes_kernel<3, 2>), grid, dim3(256), 0, st, x, cond, w, bias, out, B, S, Cout);
        return check_launch("conv_first_lanes");
    }
    const int tiles = (S + 15) / 16, L = 16 + KS - 1;
    size_t lds = ((size_t)Cin * KS * KS * Cout + (size_t)Cin * L * L) * sizeof(float);
    if (lds > 64 * 1024) { hd_set_error("conv_small_cin: LDS"); return -1; }
    dim3 grid(B * tiles * tiles);
    if (KS == 7) hipLaunchKernelGGL(conv_small_cin_kernel<7>, grid, dim3(256), lds, st, x, cond, w, bias, out, B, S, Cin, Cout);
    else hipLaunchKernelGGL(conv_small_cin_kernel<3>, grid, dim3(256), lds, st, x, cond, w, bias, out, B, S, Cin, Cout);
    return check_launch("conv_small_cin");
}

// out[p] = bias + sum_c x[p][c] * w[c]  -- final_conv 1x1 to one channel (src/hicdiff.py:343,387).
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, size_t P, int C) {
    const int l16 = threadIdx.x & 15;
    size_t p = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    float acc = 0.f;
    if (p < P) {
        for (int c = l16 * 4; c < C; c += 64) {
            float4 v = *reinterpret_cast<const float4*>(x + p * C + c);
            float4 wv = *reinterpret_cast<const float4*>(w + c);
            acc += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        }
    }
    acc += __shfl_xor(acc, 8); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 1);
    if (p < P && l16 == 0) out[p] = acc + bias[0];
}

int launch_rowdot(const float* x, const float* w, const float* bias, float* out, size_t P, int C, hipStream_t st) {
    if (C % 4) { hd_set_error("rowdot: C % 4"); return -1; }
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((P + 15) / 16)), dim3(256), 0, st, x, w, bias, out, P, C);
    return check_launch("rowdot");
}

// ------------------------------------------------------------------------------------------------
// Time embedding: SinusoidalPosEmb (src/hicdiff.py:122-134) or SR3 PositionalEncoding
// (src/hicdiff_sr3.py:155-165), then Linear -> exact GELU -> Linear (src/hicdiff.py:300-305).
// One workgroup per embedding row (Bt = 1 when every tile shares the timestep).  w1t/w3t are the
// Linear weights transposed to [in][out] so lanes read consecutive outputs.
__global__ __launch_bounds__(256) void time_mlp_kernel(const void* __restrict__ t, int t_kind, float tval, const StepParams* __restrict__ sp, int sr3, int dim, int time_dim,
                                                       const float* __restrict__ w1t, const float* __restrict__ b1,
                                                       const float* __restrict__ w3t, const float* __restrict__ b3,
                                                       float* __restrict__ temb, float* __restrict__ temb_act) {
    extern __shared__ float sh[];
    float* emb = sh;            // [dim]
    float* h1 = sh + dim;       // [time_dim]
    const int b = blockIdx.x, tid = threadIdx.x;
    // t == nullptr: every tile shares the scalar step value (sampling loops)
    if (sp) tval = sp->f[0];      // graph replay: the step's scalars live in device memory
    float tv = !t ? tval : (t_kind == 0) ? (float)reinterpret_cast<const long long*>(t)[b] : reinterpret_cast<const float*>(t)[b];
    const int half = dim / 2;
    for (int i = tid; i < dim; i += 256) {
        int k = i < half ? i : i - half;
        float f;
        if (sr3) f = expf(-9.210340371976184f * ((float)k / (float)half));
        else f = expf((float)k * -(9.210340371976184f / (float)(half - 1)));
        float a = tv * f;
        emb[i] = i < half ? sinf(a) : cosf(a);
    }
    __syncthreads();
    for (int j = tid; j < time_dim; j += 256) {
        float acc = b1[j];
        for (int i = 0; i < dim; ++i) acc += emb[i] * w1t[(size_t)i * time_dim + j];
        h1[j] = 0.5f * acc * (1.f + erff(acc * 0.70710678118654752f));
    }
    __syncthreads();
    for (int j = tid; j < time_dim; j += 256) {
        float acc = b3[j];
        for (int i = 0; i < time_dim; ++i) acc += h1[i] * w3t[(size_t)i * time_dim + j];
        temb[(size_t)b * time_dim + j] = acc;
        temb_act[(size_t)b * time_dim + j] = sr3 ? acc : silu_f(acc);
    }
}

int launch_time_mlp(const void* t, int t_kind, float tval, const StepParams* sp, int sr3, int Bt, int dim, int time_dim, const float* w1t, const float* b1,
                    const float* w3t, const float* b3, float* temb, float* temb_act, hipStream_t st) {
    size_t lds = (size_t)(dim + time_dim) * sizeof(float);
    hipLaunchKernelGGL(time_mlp_kernel, dim3(Bt), dim3(256), lds, st, t, t_kind, tval, sp, sr3, dim, time_dim, w1t, b1, w3t, b3, temb,
                       temb_act);
    return check_launch("time_mlp");
}

// All per-block FiLM projections in one launch: out[b][n] = bias[n] + sum_k act[b][k] * wt[k][n]
// (the `mlp` of every ResnetBlock, src/hicdiff.py:176-179,189-191; SR3 noise_func
// src/hicdiff_sr3.py:167-183), wt = the Linear weights transposed and concatenated along n.
__global__ __launch_bounds__(256) void film_kernel(const float* __restrict__ act, int K, const float* __restrict__ wt,
                                                   const float* __restrict__ bias, int N, float* __restrict__ out) {
    extern __shared__ float a[];
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < K; i += 256) a[i] = act[(size_t)b * K + i];
    __syncthreads();
    if (n >= N) return;
    float acc = bias[n];
    for (int k = 0; k < K; ++k) acc += a[k] * wt[(size_t)k * N + n];
    out[(size_t)b * N + n] = acc;
}

int launch_film(const float* act, int Bt, int K, const float* wt, const float* bias, int N, float* out, hipStream_t st) {
    hipLaunchKernelGGL(film_kernel, dim3((N + 255) / 256, Bt), dim3(256), K * sizeof(float), st, act, K, wt, bias, N, out);
    return check_launch("film");
}

// The sampling loops feed every tile the same step value, so the time MLP (src/hicdiff.py:286-292; SR3 src/hicdiff_sr3.py:326-334) and all
// FiLM projections have ONE input row per step.  One launch does both: every workgroup recomputes the 2-layer MLP (82 k MACs) and then its
// 256 FiLM outputs.  Each dot product is cut four ways over the waves (k quarters) with float4 columns per lane, so the serial chain is
// K/4 coalesced 16-byte loads instead of K dependent 4-byte ones (time_mlp 39 us + film 32 us -> one launch of a few us).
__global__ __launch_bounds__(256) void time_film_kernel(float tval, const StepParams* __restrict__ sp, int sr3, int dim, int time_dim,
                                                        const float* __restrict__ w1t, const float* __restrict__ b1, const float* __restrict__ w3t,
                                                        const float* __restrict__ b3, const float* __restrict__ wt, const float* __restrict__ bias, int N,
                                                        float* __restrict__ out) {
    __shared__ float emb[256], h1[1024], act[1024];
    __shared__ float4 red[4][64];
    const int tid = threadIdx.x, q = tid & 63, kg = tid >> 6;       // column quad, k quarter
    if (sp) tval = sp->f[0];
    const int half = dim / 2;
    for (int i = tid; i < dim; i += 256) {
        const int k = i < half ? i : i - half;
        const float f = sr3 ? expf(-9.210340371976184f * ((float)k / (float)half)) : expf((float)k * -(9.210340371976184f / (float)(half - 1)));
        const float a = tval * f;
        emb[i] = i < half ? sinf(a) : cosf(a);
    }
    __syncthreads();
    // y[4q..4q+3] = sum_k x[k] * W[k][4q..] over this wave's k quarter; W row-major [K][ld]
    auto gemv4 = [&](const float* x, int K, const float* W, int ld, int col0, bool live) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k0 = kg * (K / 4), k1 = k0 + K / 4;
        if (live) {
#pragma unroll 8
            for (int k = k0; k < k1; ++k) {
                const float4 w = *reinterpret_cast<const float4*>(W + (size_t)k * ld + col0);
                const float xv = x[k];
                acc.x = fmaf(xv, w.x, acc.x); acc.y = fmaf(xv, w.y, acc.y); acc.z = fmaf(xv, w.z, acc.z); acc.w = fmaf(xv, w.w, acc.w);
            }
        }
        red[kg][q] = acc;
        __syncthreads();
        float4 r = red[0][q];
        if (kg == 0) {
#pragma unroll
            for (int g = 1; g < 4; ++g) { const float4 t = red[g][q]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
        }
        __syncthreads();
        return r;                                                   // valid in wave 0
    };
    for (int c0 = 0; c0 < time_dim; c0 += 256) {                    // time_dim: 256 (UNet, dim 64) or 1024 (hicedrn)
        const float4 r = gemv4(emb, dim, w1t, time_dim, c0 + 4 * q, true);
        if (kg == 0) {
            const float v[4] = {r.x + b1[c0 + 4 * q], r.y + b1[c0 + 4 * q + 1], r.z + b1[c0 + 4 * q + 2], r.w + b1[c0 + 4 * q + 3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) h1[c0 + 4 * q + j] = 0.5f * v[j] * (1.f + erff(v[j] * 0.70710678118654752f));
        }
    }
    __syncthreads();
    for (int c0 = 0; c0 < time_dim; c0 += 256) {
        const float4 r = gemv4(h1, time_dim, w3t, time_dim, c0 + 4 * q, true);
        if (kg == 0) {
            const float v[4] = {r.x + b3[c0 + 4 * q], r.y + b3[c0 + 4 * q + 1], r.z + b3[c0 + 4 * q + 2], r.w + b3[c0 + 4 * q + 3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) act[c0 + 4 * q + j] = sr3 ? v[j] : silu_f(v[j]);
        }
    }
    __syncthreads();
    const int n0 = blockIdx.x * 256 + 4 * q;
    const float4 r = gemv4(act, time_dim, wt, N, n0, n0 < N);
    if (kg == 0 && n0 < N) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + n0);
        *reinterpret_cast<float4*>(out + n0) = make_float4(r.x + bv.x, r.y + bv.y, r.z + bv.z, r.w + bv.w);
    }
}

// uniform step value only (one time row); dim <= 256 and a multiple of 16, time_dim a multiple of 256 up to 1024, N a multiple of 4 (the caller checks)
int launch_time_film(float tval, const StepParams* sp, int sr3, int dim, int time_dim, const float* w1t, const float* b1, const float* w3t,
                     const float* b3, const float* wt, const float* bias, int N, float* out, hipStream_t st) {
    hipLaunchKernelGGL(time_film_kernel, dim3((N + 255) / 256), dim3(256), 0, st, tval, sp, sr3, dim, time_dim, w1t, b1, w3t, b3, wt, bias, N, out);
    return check_launch("time_film");
}

// ------------------------------------------------------------------------------------------------
// GroupNorm (nn.GroupNorm(8, C), src/hicdiff.py:159).  Statistics are kept as per-channel partial
// (sum, sum of squares) over pixel slots -- the same format the conv epilogue emits -- and folded
// into one per-(sample, channel) affine by gn_finalize so the consumer applies norm + FiLM + SiLU
// with a single fma + silu per element.
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int HW, int C, int chunk, int slots,
                                                         float* __restrict__ part) {
    __shared__ float sh[256][2];
    const int slot = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int CT = C < 256 ? C : 256, rows = 256 / CT;
    const int r = tid / CT, c = tid % CT;
    const int p0 = slot * chunk, p1 = min(HW, p0 + chunk);
    for (int cc = c; cc < C; cc += CT) {
        float s1 = 0.f, s2 = 0.f;
        if (r < rows)
            for (int p = p0 + r; p < p1; p += rows) {
                float v = x[((size_t)b * HW + p) * C + cc];
                s1 += v; s2 += v * v;
            }
        sh[tid][0] = s1; sh[tid][1] = s2;
        __syncthreads();
        if (r == 0) {
            for (int j = 1; j < rows; ++j) { s1 += sh[j * CT + c][0]; s2 += sh[j * CT + c][1]; }
            float* d = part + (((size_t)b * slots + slot) * C + cc) * 2;
            d[0] = s1; d[1] = s2;
        }
        __syncthreads();
    }
}

int launch_gn_partial(const float* x, int B, int HW, int C, float* part, int* slots_out, hipStream_t st) {
    const int chunk = 256;
    const int slots = (HW + chunk - 1) / chunk;
    *slots_out = slots;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(slots, B), dim3(256), 0, st, x, HW, C, chunk, slots, part);
    return check_launch("gn_partial");
}

// film_mode 0: none; 1: x*(scale+1)+shift with [scale | shift] at film[b*bs + off + {0..C, C..2C}]
// (src/hicdiff.py:166-168,191); 2: SR3 additive embedding AFTER the SiLU (src/hicdiff_sr3.py:249),
// exported as E.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, int slots, int B, int HW, int C,
                                                         int groups, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ film,
                                                         int film_bstride, int film_off, int film_mode,
                                                         float* __restrict__ A, float* __restrict__ Bv, float* __restrict__ E,
                                                         float* __restrict__ stats_out) {
    // one wave per (sample, group): lanes stride over the group's slots x channels partial sums (fixed
    // order, so the result does not depend on anything but the tiling), fp64 combine.
    const int b = blockIdx.x / groups, g = blockIdx.x % groups, lane = threadIdx.x;
    const int cg = C / groups, total = slots * cg;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < total; i += 64) {
        const int s = i / cg, j = i - s * cg;
        const float2 v = *reinterpret_cast<const float2*>(part + (((size_t)b * slots + s) * C + g * cg + j) * 2);
        s1 += v.x; s2 += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const double n = (double)HW * cg;
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    if (stats_out && lane == 0) { stats_out[2 * blockIdx.x] = (float)mean; stats_out[2 * blockIdx.x + 1] = rstd; }   // training keeps (mean, rstd) per (sample, group)
    for (int j = lane; j < cg; j += 64) {
        const int c = g * cg + j, i = b * C + c;
        float a = rstd * gamma[c];
        float bb = beta[c] - (float)mean * a;
        if (film_mode == 1) {
            const float sc = film[(size_t)b * film_bstride + film_off + c] + 1.f;
            const float sf = film[(size_t)b * film_bstride + film_off + C + c];
            a *= sc; bb = bb * sc + sf;
        } else if (film_mode == 2) {
            E[i] = film[(size_t)b * film_bstride + film_off + c];
        }
        A[i] = a; Bv[i] = bb;
    }
}

int launch_gn_finalize(const float* part, int slots, int B, int HW, int C, int groups, const float* gamma, const float* beta,
                       const float* film, int film_bstride, int film_off, int film_mode, float* A, float* Bv, float* E,
                       hipStream_t st, float* stats_out) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, st, part, slots, B, HW, C, groups, gamma,
                       beta, film, film_bstride, film_off, film_mode, A, Bv, E, stats_out);
    return check_launch("gn_finalize");
}

// out = silu(h * A[b][c] + Bv[b][c]) + res  -- ResnetBlock tail with identity shortcut (src/hicdiff.py:195-197)
__global__ __launch_bounds__(256) void affine_silu_add_kernel(const float* __restrict__ h, const float* __restrict__ A,
                                                              const float* __restrict__ Bv, const float* __restrict__ res,
                                                              float* __restrict__ out, size_t n4, int HWC4, int C4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / HWC4), c4 = (int)(i % C4);
        float4 v = reinterpret_cast<const float4*>(h)[i];
        const float4 a = reinterpret_cast<const float4*>(A)[b * C4 + c4];
        const float4 bb = reinterpret_cast<const float4*>(Bv)[b * C4 + c4];
        const float4 r = reinterpret_cast<const float4*>(res)[i];
        v.x = silu_f(v.x * a.x + bb.x) + r.x; v.y = silu_f(v.y * a.y + bb.y) + r.y;
        v.z = silu_f(v.z * a.z + bb.z) + r.z; v.w = silu_f(v.w * a.w + bb.w) + r.w;
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// The same tail that also leaves the per-pixel channel LayerNorm statistics (mean, rstd) of its OUTPUT for the attention
// block that follows (PreNorm, src/hicdiff.py:99-118): a pixel's C/4 float4 lanes are consecutive lanes of one wave.
template <int C4>
__global__ __launch_bounds__(256) void affine_silu_add_stats_kernel(const float* __restrict__ h, const float* __restrict__ A,
                                                                    const float* __restrict__ Bv, const float* __restrict__ res,
                                                                    float* __restrict__ out, float* __restrict__ stats, size_t n4, int HWC4) {
    // (n4 is a multiple of C4 and a pixel's C4 lanes are an aligned lane group: a group is inside the loop as a whole or not at all)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / HWC4), c4 = (int)(i % C4);
        float4 v = reinterpret_cast<const float4*>(h)[i];
        const float4 a = reinterpret_cast<const float4*>(A)[b * C4 + c4];
        const float4 bb = reinterpret_cast<const float4*>(Bv)[b * C4 + c4];
        const float4 r = reinterpret_cast<const float4*>(res)[i];
        v.x = silu_f(v.x * a.x + bb.x) + r.x; v.y = silu_f(v.y * a.y + bb.y) + r.y;
        v.z = silu_f(v.z * a.z + bb.z) + r.z; v.w = silu_f(v.w * a.w + bb.w) + r.w;
        reinterpret_cast<float4*>(out)[i] = v;
        float s = v.x + v.y + v.z + v.w;
#pragma unroll
        for (int m = C4 / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        const float mean = s / (4 * C4);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        float q = dx * dx + dy * dy + dz * dz + dw * dw;
#pragma unroll
        for (int m = C4 / 2; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
        if (c4 == 0) { const size_t px = i / C4; stats[2 * px] = mean; stats[2 * px + 1] = 1.f / sqrtf(q / (4 * C4) + 1e-5f); }
    }
}

// stats != nullptr: also write the LayerNorm statistics of the output (C = 64, 128 or 256: returns 1 when it did, 0 when the caller still
// has to run ln_stats; negative on error).  By the channel count only: until round 4 the rule also asked for B*HW*C/4 % 256 == 0, which made
// the summation order of the statistics -- the two kernels reduce differently -- depend on the batch (100-pixel maps of 64 channels: 4 tiles
// took this kernel, 2 or 6 the other; tools/lane_diag.py found it, tests/test_gpu_parity.py::test_batch_independence_odd_shapes keeps it).
int launch_affine_silu_add(const float* h, const float* A, const float* Bv, const float* res, float* out, int B, int HW, int C,
                           hipStream_t st, float* stats) {
    const size_t n4 = (size_t)B * HW * C / 4;
    unsigned grid = (unsigned)((n4 + 255) / 256);
    if (grid > 16384) grid = 16384;
    if (stats && (C == 64 || C == 128 || C == 256)) {
        if (C == 64) hipLaunchKernelGGL(affine_silu_add_stats_kernel<16>, dim3(grid), dim3(256), 0, st, h, A, Bv, res, out, stats, n4, HW * C / 4);
        else if (C == 128) hipLaunchKernelGGL(affine_silu_add_stats_kernel<32>, dim3(grid), dim3(256), 0, st, h, A, Bv, res, out, stats, n4, HW * C / 4);
        else hipLaunchKernelGGL(affine_silu_add_stats_kernel<64>, dim3(grid), dim3(256), 0, st, h, A, Bv, res, out, stats, n4, HW * C / 4);
        return check_launch("affine_silu_add_stats") == 0 ? 1 : -3;
    }
    hipLaunchKernelGGL(affine_silu_add_kernel, dim3(grid), dim3(256), 0, st, h, A, Bv, res, out, n4, HW * C / 4, C / 4);
    return check_launch("affine_silu_add");
}

// ------------------------------------------------------------------------------------------------
// Channel LayerNorm (src/hicdiff.py:99-108): per pixel over C, biased variance, (var+eps).rsqrt().
// 16 lanes per pixel, row kept in registers (C <= 1024).
#define LN_MAXV 16
__device__ __forceinline__ void ln_row_stats(const float* row, int C, int l16, float4 (&v)[LN_MAXV], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        int c = l16 * 4 + j * 64;
        if (c < C) { v[j] = *reinterpret_cast<const float4*>(row + c); s += v[j].x + v[j].y + v[j].z + v[j].w; }
    }
    s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
    mean = s / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        int c = l16 * 4 + j * 64;
        if (c < C) {
            float a = v[j].x - mean, b = v[j].y - mean, d = v[j].z - mean, e = v[j].w - mean;
            q += a * a + b * b + d * d + e * e;
        }
    }
    q += __shfl_xor(q, 8); q += __shfl_xor(q, 4); q += __shfl_xor(q, 2); q += __shfl_xor(q, 1);
    rstd = 1.f / sqrtf(q / C + 1e-5f);
}

__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, size_t P, int C, float* __restrict__ stats) {
    const int l16 = threadIdx.x & 15;
    size_t p = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    size_t pc = p < P ? p : P - 1;
    float4 v[LN_MAXV];
    float mean, rstd;
    ln_row_stats(x + pc * C, C, l16, v, mean, rstd);
    if (p < P && l16 == 0) { stats[2 * p] = mean; stats[2 * p + 1] = rstd; }
}

int launch_ln_stats(const float* x, size_t P, int C, float* stats, hipStream_t st) {
    if (C % 4 || C > 64 * LN_MAXV) { hd_set_error("ln_stats: unsupported C"); return -1; }
    hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)((P + 15) / 16)), dim3(256), 0, st, x, P, C, stats);
    return check_launch("ln_stats");
}

// out = LayerNorm(y) * g + res  (to_out's LayerNorm + Residual, src/hicdiff.py:207-210,64-70)
__global__ __launch_bounds__(256) void ln_residual_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                          const float* __restrict__ res, float* __restrict__ out, size_t P, int C) {
    const int l16 = threadIdx.x & 15;
    size_t p = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    size_t pc = p < P ? p : P - 1;
    float4 v[LN_MAXV];
    float mean, rstd;
    ln_row_stats(y + pc * C, C, l16, v, mean, rstd);
    if (p >= P) return;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        int c = l16 * 4 + j * 64;
        if (c < C) {
            const float4 gg = *reinterpret_cast<const float4*>(g + c);
            const float4 r = *reinterpret_cast<const float4*>(res + p * C + c);
            float4 o;
            o.x = (v[j].x - mean) * rstd * gg.x + r.x; o.y = (v[j].y - mean) * rstd * gg.y + r.y;
            o.z = (v[j].z - mean) * rstd * gg.z + r.z; o.w = (v[j].w - mean) * rstd * gg.w + r.w;
            *reinterpret_cast<float4*>(out + p * C + c) = o;
        }
    }
}

int launch_ln_residual(const float* y, const float* g, const float* res, float* out, size_t P, int C, hipStream_t st) {
    if (C % 4 || C > 64 * LN_MAXV) { hd_set_error("ln_residual: unsupported C"); return -1; }
    hipLaunchKernelGGL(ln_residual_kernel, dim3((unsigned)((P + 15) / 16)), dim3(256), 0, st, y, g, res, out, P, C);
    return check_launch("ln_residual");
}

// ------------------------------------------------------------------------------------------------
// LinearAttention (src/hicdiff.py:212-227), heads x 32.  qkv is [B][HW][3*heads*32] with q | k | v
// blocks of heads*32 channels.
//   context[d][e] = sum_n softmax_n(k[d][:])[n] * v[e][n] / HW
// Tokens are cut into SPLIT-token ranges: one workgroup per (b, head, range) produces the range's column
// maxima, exp-sums and un-normalised 32x32 context relative to its own maxima; linattn_combine rescales
// the ranges to the global maxima and normalises (flash-style two-level softmax, fixed order).
#define LINATTN_SPLIT 512
__global__ __launch_bounds__(256) void linattn_context_kernel(const float* __restrict__ qkv, int HW, int heads, int nsplit,
                                                              float* __restrict__ pmax, float* __restrict__ psum,
                                                              float* __restrict__ pctx) {
    constexpr int D = 32, CH = 64;
    __shared__ float ks[CH][D + 1];
    __shared__ float vs[CH][D];
    __shared__ float red[8][D];
    __shared__ float kmax[D];
    const int bh = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
    const int b = bh / heads, h = bh % heads, tid = threadIdx.x;
    const int C3 = 3 * heads * D;
    const float* base = qkv + (size_t)b * HW * C3;
    const int koff = heads * D + h * D, voff = 2 * heads * D + h * D;
    const int nb = sp * LINATTN_SPLIT, ne = min(HW, nb + LINATTN_SPLIT);
    const int d = tid & 31, sub = tid >> 5;   // 8 sub-rows x 32 d
    float mx = -3.0e38f;
    for (int n = nb + sub; n < ne; n += 8) mx = fmaxf(mx, base[(size_t)n * C3 + koff + d]);
    red[sub][d] = mx;
    __syncthreads();
    if (tid < D) { float m = red[0][tid]; for (int j = 1; j < 8; ++j) m = fmaxf(m, red[j][tid]); kmax[tid] = m; }
    __syncthreads();
    const int dd = tid >> 3, e0 = (tid & 7) * 4;   // thread -> (d = tid/8, four e)
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, ps = 0.f;
    const float m_d = kmax[dd];
    for (int n0 = nb; n0 < ne; n0 += CH) {
        const int cnt = min(CH, ne - n0);
        __syncthreads();
        for (int i = tid; i < CH * D; i += 256) {
            int r = i / D, c = i % D;
            float kv = 0.f, vv = 0.f;
            // the exponential is taken once per (token, d) here, not once per (token, d, e-quad) in the loop below (eight threads share a d)
            if (r < cnt) { kv = __expf(base[(size_t)(n0 + r) * C3 + koff + c] - kmax[c]); vv = base[(size_t)(n0 + r) * C3 + voff + c]; }
            ks[r][c] = kv; vs[r][c] = vv;
        }
        __syncthreads();
        for (int r = 0; r < cnt; ++r) {
            const float pexp = ks[r][dd];
            const float4 vv = *reinterpret_cast<const float4*>(&vs[r][e0]);
            ps += pexp;
            acc0 += pexp * vv.x; acc1 += pexp * vv.y; acc2 += pexp * vv.z; acc3 += pexp * vv.w;
        }
    }
    const size_t slot = (size_t)bh * nsplit + sp;
    if ((tid & 7) == 0) { pmax[slot * D + dd] = m_d; psum[slot * D + dd] = ps; }
    *reinterpret_cast<float4*>(pctx + (slot * D + dd) * D + e0) = make_float4(acc0, acc1, acc2, acc3);
}

__global__ __launch_bounds__(256) void linattn_combine_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                              const float* __restrict__ pctx, int nsplit, int HW,
                                                              float* __restrict__ ctx) {
    constexpr int D = 32;
    const int bh = blockIdx.x, tid = threadIdx.x, dd = tid >> 3, e0 = (tid & 7) * 4;
    float gm = -3.0e38f;
    for (int s = 0; s < nsplit; ++s) gm = fmaxf(gm, pmax[((size_t)bh * nsplit + s) * D + dd]);
    float sum = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const size_t slot = (size_t)bh * nsplit + s;
        const float f = __expf(pmax[slot * D + dd] - gm);
        sum += psum[slot * D + dd] * f;
        const float4 c = *reinterpret_cast<const float4*>(pctx + (slot * D + dd) * D + e0);
        a0 += c.x * f; a1 += c.y * f; a2 += c.z * f; a3 += c.w * f;
    }
    const float inv = 1.f / (sum * (float)HW);
    *reinterpret_cast<float4*>(ctx + ((size_t)bh * D + dd) * D + e0) = make_float4(a0 * inv, a1 * inv, a2 * inv, a3 * inv);
}

int launch_linattn_combine(const float* pmax, const float* psum, const float* pctx, int B, int heads, int nsplit, int HW, float* ctx,
                           hipStream_t st) {
    hipLaunchKernelGGL(linattn_combine_kernel, dim3(B * heads), dim3(256), 0, st, pmax, psum, pctx, nsplit, HW, ctx);
    return check_launch("linattn_combine");
}

size_t linattn_scratch_floats(int B, int HW, int heads) {
    const int nsplit = (HW + LINATTN_SPLIT - 1) / LINATTN_SPLIT;
    return (size_t)B * heads * nsplit * (32 + 32 + 32 * 32);
}

int launch_linattn_context(const float* qkv, int B, int HW, int heads, float* scratch, float* ctx, hipStream_t st) {
    const int nsplit = (HW + LINATTN_SPLIT - 1) / LINATTN_SPLIT;
    const size_t slots = (size_t)B * heads * nsplit;
    float* pmax = scratch; float* psum = pmax + slots * 32; float* pctx = psum + slots * 32;
    hipLaunchKernelGGL(linattn_context_kernel, dim3((unsigned)slots), dim3(256), 0, st, qkv, HW, heads, nsplit, pmax, psum, pctx);
    hipLaunchKernelGGL(linattn_combine_kernel, dim3(B * heads), dim3(256), 0, st, pmax, psum, pctx, nsplit, HW, ctx);
    return check_launch("linattn_context");
}

//   out[n][h*32+e] = scale * sum_d context[d][e] * softmax_d(q[:, n])[d]    (one thread per (pixel, head))
__global__ __launch_bounds__(256) void linattn_apply_kernel(const float* __restrict__ qkv, int qstride, const float* __restrict__ ctx,
                                                            int HW, int heads, float* __restrict__ out) {
    constexpr int D = 32;
    __shared__ float cs[D][D];
    const int chunks = (HW + 255) / 256;
    const int bh = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const int b = bh / heads, h = bh % heads, tid = threadIdx.x;
    for (int i = tid; i < D * D; i += 256) cs[i / D][i % D] = ctx[(size_t)bh * D * D + i];
    __syncthreads();
    const int n = chunk * 256 + tid;
    if (n >= HW) return;
    const float* q = qkv + ((size_t)b * HW + n) * qstride + h * D;     // qstride: 384 (qkv tensor) or 128 (q only)
    float qv[D];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < D / 4; ++j) {
        float4 t = reinterpret_cast<const float4*>(q)[j];
        qv[4 * j] = t.x; qv[4 * j + 1] = t.y; qv[4 * j + 2] = t.z; qv[4 * j + 3] = t.w;
        mx = fmaxf(mx, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) { qv[j] = __expf(qv[j] - mx); s += qv[j]; }
    const float sc = 0.17677669529663687f / s;   // dim_head ** -0.5 folded with the softmax denominator
    float* o = out + ((size_t)b * HW + n) * (heads * D) + h * D;
#pragma unroll
    for (int e0 = 0; e0 < D; e0 += 4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            const float4 c4 = *reinterpret_cast<const float4*>(&cs[dd][e0]);
            a0 += c4.x * qv[dd]; a1 += c4.y * qv[dd]; a2 += c4.z * qv[dd]; a3 += c4.w * qv[dd];
        }
        *reinterpret_cast<float4*>(o + e0) = make_float4(a0 * sc, a1 * sc, a2 * sc, a3 * sc);
    }
}

int launch_linattn_apply(const float* qkv, int qstride, const float* ctx, int B, int HW, int heads, float* out, hipStream_t st) {
    const int chunks = (HW + 255) / 256;
    hipLaunchKernelGGL(linattn_apply_kernel, dim3(B * heads * chunks), dim3(256), 0, st, qkv, qstride, ctx, HW, heads, out);
    return check_launch("linattn_apply");
}

// Full softmax attention of the mid block (src/hicdiff.py:239-251): n = HW tokens (64 at S=64,
// 25 at S=40), heads x 32.  One workgroup per (b, head, 256-query chunk); thread i owns one query and
// streams the keys/values through LDS in chunks with a running max / sum (online softmax):
// out[i][h*32+d] = sum_j softmax_j(scale * q_i . k_j) v_j[d].
__global__ __launch_bounds__(256) void attn_full_kernel(const float* __restrict__ qkv, int HW, int heads, float* __restrict__ out) {
    constexpr int D = 32, CH = 64;
    __shared__ float ks[CH][D + 1];
    __shared__ float vs[CH][D];
    const int qchunks = (HW + 255) / 256;
    const int bh = blockIdx.x / qchunks, qc = blockIdx.x % qchunks;
    const int b = bh / heads, h = bh % heads, tid = threadIdx.x;
    const int C3 = 3 * heads * D;
    const float* base = qkv + (size_t)b * HW * C3;
    const int qi = qc * 256 + tid;
    const bool active = qi < HW;
    float q[D], acc[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { q[j] = active ? base[(size_t)qi * C3 + h * D + j] * 0.17677669529663687f : 0.f; acc[j] = 0.f; }
    float mx = -3.0e38f, den = 0.f;
    for (int n0 = 0; n0 < HW; n0 += CH) {
        const int cnt = min(CH, HW - n0);
        __syncthreads();
        for (int i = tid; i < CH * D; i += 256) {
            int r = i / D, c = i % D;
            float kv = 0.f, vv = 0.f;
            if (r < cnt) { kv = base[(size_t)(n0 + r) * C3 + heads * D + h * D + c]; vv = base[(size_t)(n0 + r) * C3 + 2 * heads * D + h * D + c]; }
            ks[r][c] = kv; vs[r][c] = vv;
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) sc += q[d] * ks[j][d];
            if (sc > mx) {                      // rescale the running sums to the new maximum
                const float f = __expf(mx - sc);
                den *= f;
#pragma unroll
                for (int d = 0; d < D; ++d) acc[d] *= f;
                mx = sc;
            }
            const float pe = __expf(sc - mx);
            den += pe;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] += pe * vs[j][d];
        }
    }
    if (!active) return;
    const float inv = 1.f / den;
    float* o = out + ((size_t)b * HW + qi) * (heads * D) + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) *reinterpret_cast<float4*>(o + d) = make_float4(acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv);
}

int launch_attn_full_mfma(const float* qkv, int B, int HW, int heads, float* out, hipStream_t st);   // attn_mfma.hip; returns 1 when the shape is not its

int launch_attn_full(const float* qkv, int B, int HW, int heads, float* out, hipStream_t st) {
    static const bool no_mfma = getenv("HICDIFF_ATTN_SCALAR") != nullptr;
    if (!no_mfma) {
        const int rc = launch_attn_full_mfma(qkv, B, HW, heads, out, st);
        if (rc <= 0) return rc;
    }
    const int qchunks = (HW + 255) / 256;
    hipLaunchKernelGGL(attn_full_kernel, dim3(B * heads * qchunks), dim3(256), 0, st, qkv, HW, heads, out);
    return check_launch("attn_full");
}

// ------------------------------------------------------------------------------------------------
// Device Gaussian generator for perf runs: Philox4x32-10 keyed by seed, counter =
// (pixel quad, global tile index, step, stream), Box-Muller.  Independent of the rank count because
// the key material is the GLOBAL tile index.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}

__device__ __forceinline__ float4 philox_normal4(uint64_t seed, uint32_t quad, uint64_t tile, uint32_t step, uint32_t stream) {
    uint32_t c[4] = {quad, (uint32_t)tile, (uint32_t)(tile >> 32) ^ (stream << 28), step};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const float u0 = ((float)(c[0] >> 8) + 0.5f) * (1.f / 16777216.f), u1 = ((float)(c[1] >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u2 = ((float)(c[2] >> 8) + 0.5f) * (1.f / 16777216.f), u3 = ((float)(c[3] >> 8) + 0.5f) * (1.f / 16777216.f);
    const float r0 = sqrtf(-2.f * __logf(u0)), r1 = sqrtf(-2.f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, int B, int SS4, uint64_t seed, uint64_t tile_off,
                                                    uint32_t step, uint32_t nstream) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * SS4) return;
    const int b = (int)(i / SS4), q = (int)(i % SS4);
    reinterpret_cast<float4*>(out)[i] = philox_normal4(seed, q, tile_off + b, step, nstream);
}

int launch_randn(float* out, int B, int S, uint64_t seed, uint64_t tile_off, uint32_t step, hipStream_t st, uint32_t nstream) {
    const int SS4 = S * S / 4;
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)(((size_t)B * SS4 + 255) / 256)), dim3(256), 0, st, out, B, SS4, seed, tile_off,
                       step, nstream);
    return check_launch("randn");
}

// p_sample tail (src/hicdiff.py:529-533,553-560,589,599-600), four pixels per thread.
__global__ __launch_bounds__(256) void ddpm_update_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                          const float* __restrict__ noise, float c_recip, float c_recipm1,
                                                          float coef1, float coef2, float sigma, float* __restrict__ x0_out, int B,
                                                          int SS4, uint64_t seed, uint64_t tile_off, uint32_t step,
                                                          const StepParams* __restrict__ sp, float coef_eps, uint32_t tile_add) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * SS4) return;
    if (sp) {
        c_recip = sp->f[1]; c_recipm1 = sp->f[2]; coef1 = sp->f[3]; coef2 = sp->f[4]; sigma = sp->f[5]; coef_eps = sp->f[6];
        seed = sp->seed; tile_off = sp->tile_off + tile_add; step = sp->step;      // tile_add: first tile of this launch inside the call's batch (chained steps)
    }
    const float4 xv = reinterpret_cast<float4*>(x)[i];
    const float4 ev = reinterpret_cast<const float4*>(eps)[i];
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sigma != 0.f) {
        if (noise) z = reinterpret_cast<const float4*>(noise)[i];
        else z = philox_normal4(seed, (uint32_t)(i % SS4), tile_off + i / SS4, step, 0);
    }
    float4 x0, o;
#define HD_STEP(f)                                                           \
    x0.f = fminf(fmaxf(c_recip * xv.f - c_recipm1 * ev.f, -1.f), 1.f);      \
    o.f = coef1 * x0.f + coef2 * xv.f + coef_eps * ev.f + sigma * z.f;
    HD_STEP(x) HD_STEP(y) HD_STEP(z) HD_STEP(w)
#undef HD_STEP
    reinterpret_cast<float4*>(x)[i] = o;
    if (x0_out) reinterpret_cast<float4*>(x0_out)[i] = x0;
}

int launch_ddpm_update(float* x, const float* eps, const float* noise, float c_recip, float c_recipm1, float coef1, float coef2,
                       float sigma, float* x0_out, int B, int S, uint64_t seed, uint64_t tile_off, uint32_t step, const StepParams* sp,
                       hipStream_t st, float coef_eps, uint32_t tile_add) {
    if ((S * S) % 4) { hd_set_error("tile size must make S*S a multiple of 4"); return -1; }
    const int SS4 = S * S / 4;
    hipLaunchKernelGGL(ddpm_update_kernel, dim3((unsigned)(((size_t)B * SS4 + 255) / 256)), dim3(256), 0, st, x, eps, noise, c_recip,
                       c_recipm1, coef1, coef2, sigma, x0_out, B, SS4, seed, tile_off, step, sp, coef_eps, tile_add);
    return check_launch("ddpm_update");
}

// DDRM step for the identity degradation (src/functions/denoising.py:66-104): every pixel takes the
// same branch because all singular values are 1.
__global__ __launch_bounds__(256) void ddrm_update_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                          const float* __restrict__ y, const float* __restrict__ z, float sqrt_at,
                                                          float sqrt_1m_at, float sqrt_at_next, float sigma_next, float sigma_0,
                                                          float etaA, float etaB, float etaC, float* __restrict__ x0_out,
                                                          size_t n, int SS, uint64_t seed, uint64_t tile_off, uint32_t step,
                                                          const StepParams* __restrict__ sp, uint32_t tile_add) {
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 * 4 >= n) return;
    if (sp) {
        sqrt_at = sp->f[1]; sqrt_1m_at = sp->f[2]; sqrt_at_next = sp->f[3]; sigma_next = sp->f[4]; sigma_0 = sp->f[5];
        etaA = sp->f[6]; etaB = sp->f[7]; etaC = sp->f[8];
        seed = sp->seed; tile_off = sp->tile_off + tile_add; step = sp->step;      // tile_add: first tile of this launch inside the call's batch (chained steps)
    }
    const float4 xv = reinterpret_cast<float4*>(x)[i4];
    const float4 ev = reinterpret_cast<const float4*>(eps)[i4];
    const float4 yv = reinterpret_cast<const float4*>(y)[i4];
    const bool before = sigma_next > sigma_0, after = sigma_next < sigma_0;
    const int stream = before ? 2 : (after ? 1 : 0);
    float4 zv;
    if (z) zv = reinterpret_cast<const float4*>(z + (size_t)stream * n)[i4];
    else zv = philox_normal4(seed, (uint32_t)((i4 * 4 % SS) / 4), tile_off + (i4 * 4) / SS, step, stream);
    const float std_c = sigma_next * etaC, til_c = sqrtf(sigma_next * sigma_next - std_c * std_c);
    const float std_a = sigma_next * etaA, til_a = sqrtf(sigma_next * sigma_next - std_a * std_a);
    const float diff_b = before ? sqrtf(sigma_next * sigma_next - sigma_0 * sigma_0 * etaB * etaB) : 0.f;
    float4 x0, o;
#define HD_STEP(f)                                                                          \
    x0.f = (xv.f - ev.f * sqrt_1m_at) / sqrt_at;                                            \
    {                                                                                       \
        float nx;                                                                           \
        if (before) nx = yv.f * etaB + (1.f - etaB) * x0.f + diff_b * zv.f;                 \
        else if (after) nx = x0.f + til_a * ((yv.f - x0.f) / sigma_0) + std_a * zv.f;       \
        else nx = x0.f + til_c * ev.f + std_c * zv.f;                                       \
        o.f = sqrt_at_next * nx;                                                            \
    }
    HD_STEP(x) HD_STEP(y) HD_STEP(z) HD_STEP(w)
#undef HD_STEP
    reinterpret_cast<float4*>(x)[i4] = o;
    if (x0_out) reinterpret_cast<float4*>(x0_out)[i4] = x0;
}

int launch_ddrm_update(float* x, const float* eps, const float* y, const float* z, float sqrt_at, float sqrt_1m_at,
                       float sqrt_at_next, float sigma_next, float sigma_0, float etaA, float etaB, float etaC, float* x0_out, int B,
                       int S, uint64_t seed, uint64_t tile_off, uint32_t step, const StepParams* sp, hipStream_t st, uint32_t tile_add) {
    if ((S * S) % 4) { hd_set_error("tile size must make S*S a multiple of 4"); return -1; }
    const size_t n = (size_t)B * S * S;
    hipLaunchKernelGGL(ddrm_update_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, x, eps, y, z, sqrt_at, sqrt_1m_at,
                       sqrt_at_next, sigma_next, sigma_0, etaA, etaB, etaC, x0_out, n, S * S, seed, tile_off, step, sp, tile_add);
    return check_launch("ddrm_update");
}

__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const float* __restrict__ a, const float* __restrict__ s,
                                                       float* __restrict__ out, int B, int SS) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * SS) return;
    const int b = (int)(i / SS);
    out[i] = a[b] * x0[i] + s[b] * noise[i];
}

int launch_q_sample(const float* x0, const float* noise, const float* a, const float* s, float* out, int B, int S, hipStream_t st) {
    const size_t n = (size_t)B * S * S;
    hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x0, noise, a, s, out, B, S * S);
    return check_launch("q_sample");
}

__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ pred, const float* __restrict__ target, int l2,
                                                   float* __restrict__ out, int SS) {
    __shared__ float red[256];
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < SS; i += 256) {
        float d = pred[(size_t)b * SS + i] - target[(size_t)b * SS + i];
        acc += l2 ? d * d : fabsf(d);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) out[b] = red[0] / SS;
}

int launch_loss(const float* pred, const float* target, int l2, float* out, int B, int S, hipStream_t st) {
    hipLaunchKernelGGL(loss_kernel, dim3(B), dim3(256), 0, st, pred, target, l2, out, S * S);
    return check_launch("loss");
}

// Writes one step's scalars into device memory ahead of a graph replay (kernel arguments are copied at
// enqueue time, so the host may run arbitrarily far ahead of the device).
__global__ void set_step_params_kernel(StepParams* dst, StepParams v) { *dst = v; }

int launch_set_step_params(StepParams* dst, const StepParams& v, hipStream_t st) {
    hipLaunchKernelGGL(set_step_params_kernel, dim3(1), dim3(1), 0, st, dst, v);
    return check_launch("set_step_params");
}

// Holds a stream for `us` microseconds with one idle wave (lane 1's start offset inside a chain bracket, engine.hip).  The wave sleeps
// between reads of the 100 MHz wall clock and leaves after a bounded number of reads whatever the clock says.
__global__ void spin_us_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 22); ++i) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

int launch_spin_us(int us, hipStream_t st) {
    hipLaunchKernelGGL(spin_us_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)us * 100ull);
    return check_launch("spin_us");
}

// ---- tile-quality metrics (src/Utils/loss/SSIM.py:17-37, src/Utils/stard_metrics.py:146-160) --------------------
// One workgroup per tile.  Both tiles are staged in LDS (optionally mapped from [-1,1] to [0,1] with a clamp:
// inverse_data_transform('rescaled'), src/datasets/__init__.py:214-223); every thread walks its pixels, forms the five
// 11x11 Gaussian-window sums (zero padding, fp32, row-major like a direct convolution) and the SSIM value, and
// accumulates seven sums in double.  partial[b][8] = {sum (p-t)^2, sum ssim, sum t, sum p, sum t^2, sum p^2, sum p*t, S*S};
// a second single-workgroup kernel adds the tiles up in index order (deterministic).
__constant__ unsigned int kSsimWindowBits[11] = {0x3a86cab6u, 0x3bf8ff01u, 0x3d13758cu, 0x3ddff87fu, 0x3e5a1e1fu, 0x3e8832b0u,
                                                0x3e5a1e1fu, 0x3ddff87fu, 0x3d13758cu, 0x3bf8ff01u, 0x3a86cab6u};   // gaussian(11, 1.5) as float32

__global__ __launch_bounds__(256) void tile_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ target, int S,
                                                           int rescale, double* __restrict__ partial) {
    extern __shared__ float tm_lds[];
    float* xs = tm_lds;            // pred   [S][S]
    float* ys = tm_lds + S * S;    // target [S][S]
    __shared__ float w1[11];
    __shared__ double red[4][8];
    const int b = blockIdx.x, tid = threadIdx.x, n = S * S;
    if (tid < 11) w1[tid] = __builtin_bit_cast(float, kSsimWindowBits[tid]);
    for (int i = tid; i < n; i += 256) {
        float x = pred[(size_t)b * n + i], y = target[(size_t)b * n + i];
        if (rescale) { x = fminf(fmaxf((x + 1.0f) / 2.0f, 0.0f), 1.0f); y = fminf(fmaxf((y + 1.0f) / 2.0f, 0.0f), 1.0f); }
        xs[i] = x; ys[i] = y;
    }
    __syncthreads();
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < n; i += 256) {
        const int py = i / S, px = i - py * S;
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
        for (int dy = 0; dy < 11; ++dy) {
            const int yy = py + dy - 5;
            if (yy < 0 || yy >= S) continue;
            for (int dx = 0; dx < 11; ++dx) {
                const int xx = px + dx - 5;
                if (xx < 0 || xx >= S) continue;
                const float w = w1[dy] * w1[dx];
                const float a = xs[yy * S + xx], c = ys[yy * S + xx];
                mu1 += w * a; mu2 += w * c; e11 += w * (a * a); e22 += w * (c * c); e12 += w * (a * c);
            }
        }
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu1_mu2;
        const float C1 = 0.0001f, C2 = 0.0009f;
        const float v = ((2.f * mu1_mu2 + C1) * (2.f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
        const float x = xs[i], y = ys[i], d = x - y;
        acc[0] += (double)(d * d); acc[1] += (double)v; acc[2] += (double)y; acc[3] += (double)x;
        acc[4] += (double)y * y; acc[5] += (double)x * x; acc[6] += (double)x * y;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        double v = acc[k];
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((tid & 63) == 0) red[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < 7) partial[(size_t)b * 8 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    if (tid == 7) partial[(size_t)b * 8 + 7] = (double)n;
}

__global__ __launch_bounds__(64) void tile_metrics_reduce_kernel(const double* __restrict__ partial, int B, int S, double* __restrict__ sums,
                                                                 float* __restrict__ ssim_each) {
    const int tid = threadIdx.x;
    if (tid < 8) {
        double a = 0.0;
        for (int b = 0; b < B; ++b) a += partial[(size_t)b * 8 + tid];
        sums[tid] = a;
    }
    if (ssim_each) for (int b = tid; b < B; b += 64) ssim_each[b] = (float)(partial[(size_t)b * 8 + 1] / (double)(S * S));
}

int launch_tile_metrics(const float* pred, const float* target, int B, int S, int rescale, double* partial, double* sums, float* ssim_each,
                        hipStream_t st) {
    if (S < 1 || S > 128) { hd_set_error("tile_metrics: tile size must be 1..128"); return -1; }
    hipLaunchKernelGGL(tile_metrics_kernel, dim3(B), dim3(256), (size_t)2 * S * S * sizeof(float), st, pred, target, S, rescale, partial);
    hipLaunchKernelGGL(tile_metrics_reduce_kernel, dim3(1), dim3(64), 0, st, partial, B, S, sums, ssim_each);
    return check_launch("tile_metrics");
}
