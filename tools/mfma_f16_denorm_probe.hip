// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  (round 4: the two-product arithmetic xh (wh + wl) stores wl = w - fp16(w), which is
// subnormal in fp16 for |w| < ~0.1.)  A = ones, B = 2^-20 (subnormal: the smallest normal fp16 is 2^-14): every output must be 16 * 2^-20.
//   hipcc --offload-arch=gfx950 tools/mfma_f16_denorm_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, float bval) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)bval; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    f32x4 d;
    for (int i = 0; i < 4; ++i) d[i] = 0.f;
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)b[0]; }
}
int main() {
    float* d; hipMalloc(&d, 12);
    for (float v : {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24: smallest subnormal */, 6.103515625e-05f /* 2^-14: smallest normal */}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, v);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("b = %.6e (as fp16 %.6e): 32x32x16 -> %.6e (expect %.6e), 16x16x32 -> %.6e (expect %.6e)\n", v, h[2], h[0], 16 * v, h[1], 32 * v);
    }
    return 0;
}
