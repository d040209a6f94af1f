// Stand-alone probe for the "exact zeros in the low lane of a packed multiply that broadcasts from the ODD register of a
// freshly loaded pair" observation (conv_bf16x3_kernel.h, LayerNorm loader).
//
//   hipcc -O3 --offload-arch=gfx950 tools/pk_mul_opsel_repro.hip -o /tmp/pk_repro && /tmp/pk_repro [variant] [workgroups]
//
// Every thread loads a (mean, rstd) pair with ONE global_load_dwordx2 and computes (x0 - mean) * rstd, (x1 - mean) * rstd with
// a packed multiply whose two lanes both take rstd from the pair's odd register (v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,1],
// written in inline asm so the compiler cannot choose another form), next to LDS traffic and a barrier as in the loader.
// variant 0: pair used in place (the suspect form)        variant 1: rstd copied out first with v_mov_b32 (the shipped workaround)
// variant 2: in place, s_nop 4 between the wait and the multiply        variant 3: in place, the pair loaded by two dword loads
// The host checks every output against the scalar formula and reports mismatches per lane quarter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int V>
__global__ __launch_bounds__(256) void probe(const float2* __restrict__ stats, const float2* __restrict__ x, float2* __restrict__ out, int n) {
    __shared__ float4 lds[256 * 4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ii = i < n ? i : n - 1;
    // some LDS traffic around the load, as the loader has (parameter table writes + a barrier)
    for (int k = 0; k < 4; ++k) lds[threadIdx.x * 4 + k] = make_float4(threadIdx.x, k, ii, 1.f);
    const float2 xv = x[ii];
    float2 r;
    if (V == 3) {
        float mean, rstd;
        const float* s = reinterpret_cast<const float*>(stats + ii);
        asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %2, off offset:4\n\ts_waitcnt vmcnt(0)" : "=&v"(mean), "=&v"(rstd) : "v"(s) : "memory");
        r = make_float2((xv.x - mean) * rstd, (xv.y - mean) * rstd);
    } else {
        float2 st;
        asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(st) : "v"(stats + ii) : "memory");
        const float2 d = make_float2(xv.x - st.x, xv.y - st.x);
        if (V == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(d), "v"(st));
        } else if (V == 2) {
            asm volatile("s_nop 4\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(d), "v"(st));
        } else {
            float rs;
            asm volatile("v_mov_b32 %0, %1" : "=v"(rs) : "v"(st.y));
            r = make_float2(d.x * rs, d.y * rs);
        }
    }
    __syncthreads();
    const float4 t = lds[((threadIdx.x + 17) & 255) * 4 + 1];
    if (i < n) out[i] = make_float2(r.x + 0.f * t.x, r.y);
}

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0, wgs = argc > 2 ? atoi(argv[2]) : 16384, reps = 20;
    const int n = wgs * 256;
    std::vector<float2> hs(n), hx(n), ho(n);
    for (int i = 0; i < n; ++i) {
        hs[i] = make_float2(0.25f + (i % 97) * 0.01f, 1.5f + (i % 31) * 0.125f);
        hx[i] = make_float2(1.f + (i % 13), -2.f + (i % 7) * 0.5f);
    }
    float2 *ds, *dx, *dout;
    hipMalloc(&ds, n * sizeof(float2)); hipMalloc(&dx, n * sizeof(float2)); hipMalloc(&dout, n * sizeof(float2));
    hipMemcpy(ds, hs.data(), n * sizeof(float2), hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), n * sizeof(float2), hipMemcpyHostToDevice);
    long bad[4] = {0, 0, 0, 0}, zeros_lo = 0;
    for (int rep = 0; rep < reps; ++rep) {
        hipMemset(dout, 0xff, n * sizeof(float2));
        switch (variant) {
            case 0: hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(256), 0, 0, ds, dx, dout, n); break;
            case 1: hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(256), 0, 0, ds, dx, dout, n); break;
            case 2: hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(256), 0, 0, ds, dx, dout, n); break;
            default: hipLaunchKernelGGL(probe<3>, dim3(wgs), dim3(256), 0, 0, ds, dx, dout, n); break;
        }
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        hipMemcpy(ho.data(), dout, n * sizeof(float2), hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) {
            const float e0 = (hx[i].x - hs[i].x) * hs[i].y, e1 = (hx[i].y - hs[i].x) * hs[i].y;
            if (ho[i].x != e0 || ho[i].y != e1) {
                ++bad[(i & 63) >> 4];
                if (ho[i].x == 0.f && e0 != 0.f) ++zeros_lo;
            }
        }
    }
    printf("variant %d, %d workgroups x %d launches: mismatches by lane quarter [0-15 16-31 32-47 48-63] = %ld %ld %ld %ld, exact-zero low lanes %ld\n", variant,
           wgs, reps, bad[0], bad[1], bad[2], bad[3], zeros_lo);
    return (bad[0] + bad[1] + bad[2] + bad[3]) ? 1 : 0;
}
