// Reduced form of the asymmetry in DESIGN.md section 8: the GroupNorm-sum loop of conv_epilogue (conv_device.h) on a synthetic staged tile.
// 256 threads = 32 channel quads x 8 row groups; two rounds of 64 staged rows x 128 channels; rows 0-31 of a round belong to image 0
// (sums [0]), rows 32-63 to image 1 (sums [1]).  Form A: the product's single loop with an exec-masked region; form B: a uniform branch on
// `two` plus selects (the HD_EPI_V9 block).  Both add the same values in the same order; the program reports every (thread, set, j)
// where they differ.  -DFMA=1 writes the square-accumulate as an explicit fmaf in both forms (one rounding whatever the instruction selection).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/epilogue_sum_repro.hip -o /tmp/esr && /tmp/esr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef FMA
#define FMA 0
#endif
__device__ __forceinline__ float sq_acc(float x, float s) { return FMA ? __builtin_fmaf(x, x, s) : s + x * x; }
constexpr int BN = 128, EP = BN + 4, CQ = BN / 4, NT = 256, RPP = NT / CQ, RB = 64, NPASS = RB / RPP, TM = 2;

template <int FORM>
__global__ __launch_bounds__(256) void k(const float* __restrict__ tile, const int* __restrict__ rowpix_g, const float* __restrict__ bias_g, int two_i, int Cout,
                                        float* __restrict__ out, float* __restrict__ sums) {
    __shared__ float stage[RB * EP];
    __shared__ int rowpix[128];
    const int tid = threadIdx.x, cq = tid % CQ, rg = tid / CQ, n = cq * 4;
    if (tid < 128) rowpix[tid] = rowpix_g[blockIdx.x * 128 + tid];
    const float4 b4 = *reinterpret_cast<const float4*>(bias_g + n);
    const float bias[4] = {b4.x, b4.y, b4.z, b4.w};
    float s1[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, s2[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const bool two = two_i == 2;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        __syncthreads();
        for (int i = tid; i < RB * BN; i += 256) stage[(i / BN) * EP + i % BN] = tile[((size_t)blockIdx.x * TM + tm) * RB * BN + i];
        __syncthreads();
        int pixs[NPASS];
        float4 rows[NPASS];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int lr = pass * RPP + rg;
            pixs[pass] = rowpix[(lr >> 5) * 32 * TM + tm * 32 + (lr & 31)];
            rows[pass] = *reinterpret_cast<const float4*>(stage + lr * EP + cq * 4);
        }
        if constexpr (FORM == 0) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int lr = pass * RPP + rg;
                const int pix = pixs[pass];
                const bool up = two && lr >= 32;
                const float4 a4 = rows[pass];
                const f32x4 o4 = {a4.x + bias[0], a4.y + bias[1], a4.z + bias[2], a4.w + bias[3]};
                if (pix >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = o4[j], lo = up ? 0.f : x, hi = up ? x : 0.f;
                        s1[0][j] += lo; s2[0][j] = sq_acc(lo, s2[0][j]); s1[1][j] += hi; s2[1][j] = sq_acc(hi, s2[1][j]);
                    }
                    *reinterpret_cast<f32x4*>(out + (size_t)pix * Cout + n) = o4;
                }
            }
        } else {
            if (!two) {
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    const int pix = pixs[pass];
                    const float4 a4 = rows[pass];
                    const f32x4 o4 = {a4.x + bias[0], a4.y + bias[1], a4.z + bias[2], a4.w + bias[3]};
                    if (pix >= 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float x = o4[j]; s1[0][j] += x; s2[0][j] = sq_acc(x, s2[0][j]); }
                        *reinterpret_cast<f32x4*>(out + (size_t)pix * Cout + n) = o4;
                    }
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
                    for (int j = 0; j < 4; ++j) {
                        const float x = live ? o4[j] : 0.f, lo = up ? 0.f : x, hi = up ? x : 0.f;
                        s1[0][j] += lo; s2[0][j] = sq_acc(lo, s2[0][j]); s1[1][j] += hi; s2[1][j] = sq_acc(hi, s2[1][j]);
                    }
                    if (live) *reinterpret_cast<f32x4*>(out + (size_t)pix * Cout + n) = o4;
                }
            }
        }
    }
    float* d = sums + ((size_t)blockIdx.x * 256 + tid) * 16;
    for (int h = 0; h < 2; ++h)
        for (int j = 0; j < 4; ++j) { d[h * 8 + j] = s1[h][j]; d[h * 8 + 4 + j] = s2[h][j]; }
}

int main() {
    const int NWG = 2048, Cout = 128;
    std::vector<float> tile((size_t)NWG * TM * RB * BN), bias(BN);
    std::vector<int> rowpix((size_t)NWG * 128);
    srand(1);
    for (auto& v : tile) v = (float)rand() / RAND_MAX * 4.f - 2.f;
    for (auto& v : bias) v = (float)rand() / RAND_MAX - 0.5f;
    for (int w = 0; w < NWG; ++w)
        for (int r = 0; r < 128; ++r) rowpix[(size_t)w * 128 + r] = (w % 7 == 3 && r >= 64) ? -1 : w * 128 + r;   // some tiles with a missing second image
    float *dt, *db, *dout, *ds[2];
    int* dr;
    hipMalloc(&dt, tile.size() * 4); hipMalloc(&db, bias.size() * 4); hipMalloc(&dr, rowpix.size() * 4);
    hipMalloc(&dout, (size_t)NWG * 128 * Cout * 4);
    for (int f = 0; f < 2; ++f) hipMalloc(&ds[f], (size_t)NWG * 256 * 16 * 4);
    hipMemcpy(dt, tile.data(), tile.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dr, rowpix.data(), rowpix.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<0>, dim3(NWG), dim3(256), 0, 0, dt, dr, db, 2, Cout, dout, ds[0]);
    hipLaunchKernelGGL(k<1>, dim3(NWG), dim3(256), 0, 0, dt, dr, db, 2, Cout, dout, ds[1]);
    std::vector<float> h0((size_t)NWG * 256 * 16), h1(h0.size());
    hipMemcpy(h0.data(), ds[0], h0.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h1.data(), ds[1], h1.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, badlo = 0, badhi = 0;
    for (size_t i = 0; i < h0.size(); ++i)
        if (h0[i] != h1[i]) { ++bad; if ((i % 16) < 8) ++badlo; else ++badhi; if (bad <= 5) printf("  differ at thread %zu slot %zu: %.9g vs %.9g\n", i / 16, i % 16, h0[i], h1[i]); }
    printf("form A (exec-masked single loop) vs form B (uniform branch + selects): %zu of %zu sums differ (%zu in the lower-half set, %zu in the upper-half set)\n", bad,
           h0.size(), badlo, badhi);
    return 0;
}
