// Microbenchmark: HBM store rate of the conv epilogue's access pattern (NHWC tile rows, 16 B per lane)
// versus a plain linear stream, same bytes.   hipcc --offload-arch=gfx950 -O3 store_pattern_bench.hip -o spb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// tile: 16 x 16 pixels x C channels of image [B][64][64][C]; thread -> 4 channels (cq), rows rg + 16*pass
template <int C>
__global__ __launch_bounds__(256) void tile_store(float* out, int lds_bytes_dummy) {
    extern __shared__ char smem[];
    constexpr int CQ = C / 4, RPP = 256 / CQ, NP = 256 / RPP;
    const int tile = blockIdx.x, b = tile / 16, ty = (tile % 16) / 4, tx = tile % 4;
    const int cq = threadIdx.x % CQ, rg = threadIdx.x / CQ;
    if (lds_bytes_dummy < 0) smem[threadIdx.x] = 1;
    f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
#pragma unroll 4
    for (int pass = 0; pass < NP; ++pass) {
        const int m = pass * RPP + rg, py = m / 16, px = m % 16;
        const size_t pix = ((size_t)b * 64 + ty * 16 + py) * 64 + tx * 16 + px;
        *reinterpret_cast<f32x4*>(out + pix * C + cq * 4) = v;
    }
}
__global__ __launch_bounds__(256) void linear_store(float* out, size_t n4) {
    f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) reinterpret_cast<f32x4*>(out)[i] = v;
}
int main() {
    const int B = 256, C = 64;
    const size_t n = (size_t)B * 4096 * C;
    float* out; hipMalloc(&out, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lds : {0, 70 * 1024}) {
        hipFuncSetAttribute((const void*)tile_store<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(tile_store<C>, dim3(B * 16), dim3(256), lds, 0, out, lds);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("tile_store lds=%6d: %.1f us  %.2f TB/s\n", lds, ms * 1e3, n * 4 / (ms * 1e-3) / 1e12);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(linear_store, dim3(2048), dim3(256), 0, 0, out, n / 4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("linear_store: %.1f us  %.2f TB/s\n", ms * 1e3, n * 4 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
