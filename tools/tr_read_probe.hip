// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in which lane / register.  Build: hipcc --offload-arch=gfx950 -O2
// tools/tr_read_probe.hip -o gpurun_out/tr_probe; run on the GPU box.  LDS image [pixel][96 shorts], value = pixel * 256 + channel.
// Each lane 4q + p of a 16-lane group supplies the address of row q (pixel), columns 4p .. 4p+3 of a 4 x 16 block.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 96];
    for (int i = threadIdx.x; i < 64 * 96; i += 64) lds[i] = (short)((i / 96) * 256 + (i % 96));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const short* a = lds + (8 * (g >> 1) + q) * 96 + 16 * (g & 1) + 4 * p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    *reinterpret_cast<s16x4*>(out + l * 4) = v;
}
int main() {
    short* d; short h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int pix = h[l * 4 + j] >> 8, ch = h[l * 4 + j] & 255;
            printf(" (px %2d ch %2d)", pix, ch);
            if (pix != 8 * (g >> 1) + j || ch != 16 * (g & 1) + i) ++bad;
        }
        printf("\n");
    }
    printf("expected mapping (lane i of group g: channel 16 (g & 1) + i, element j = pixel 8 (g >> 1) + j): %s\n", bad ? "MISMATCH" : "ok");
    return 0;
}
