#!/bin/bash
# Bisect of the exact-zero-lane hazard of the LayerNorm loader (log: profiles/r03_h_ln_zero_lane_bisect.txt).
#
# The fault only reproduces on the tree that introduced the workaround (commit 96daf9e, round 1) with the two `v_mov_b32` copies of the
# (mean, rstd) pair removed; neither the round-2 final tree nor the current one fails in that form.  Recipe (CPU container, then GPU box):
#
#   tools/ln_zero_lane_bisect.sh trees            # git-extracts hicdiff_amd/csrc + include of 96daf9e into tools/_r1 and builds the variants
#   gpurun -- tools/ln_zero_lane_bisect.sh run    # tools/ln_zero_lane_probe.py over every variant (two passes)
#
# Variants (conv_bf16x3_kernel.h of that tree; only the two bf16x3 translation units are rebuilt):
#   A  as committed (v_mov copies)                              B  statistics used in place (fails)
#   C  B + -mllvm -amdgpu-waitcnt-forcezero (clean)             D  B + s_waitcnt vmcnt(0) lgkmcnt(0) right behind the statistics loads
#   E  B + s_nop 1 / G  B + 2 x s_nop 7 behind the LDS stores of x_stage (store-data registers named as operands)
#   H  B + vmcnt(0) / I  B + lgkmcnt(0) at the top of x_stage;  J  B + both in front of the bf16 split
#   K  B + full wait + sched_barrier in front of every MFMA cluster;  L  B + full wait, barrier, full wait in front of the epilogue
#   M  B + lgkmcnt(0) + s_sleep 1 in front of every barrier;   N  B + every barrier doubled
set -u
cd "$(dirname "$0")/.."
R=tools/_r1
variant() {   # name, python expression turning the committed header text `s` into the variant, extra CXXFLAGS
    python3 - "$1" "$2" <<'PY'
import sys
name, expr = sys.argv[1], sys.argv[2]
s = open("/tmp/r1_orig.h").read()
VMOV = '''            asm volatile("v_mov_b32 %0, %1" : "=v"(ln_rs[j]) : "v"(st.y));
            asm volatile("v_mov_b32 %0, %1" : "=v"(ln_mu[j]) : "v"(st.x));'''
INPLACE = "            ln_rs[j] = st.y; ln_mu[j] = st.x;"
STORES = '''        *reinterpret_cast<uint4*>(d) = hi;
        *reinterpret_cast<uint4*>(d + 2 * CK) = lo;
    };'''
XTOP = '''        const char* pt = ptab + (c & 1) * pt_stride + it_pt[j];
        const float4 v0 = xform4<MODE>(xr[j][0], pt, ptv, ln_mu[j], ln_rs[j]);'''
SPLIT = '''        uint4 hi, lo;
        split8(v0, v1, hi, lo);
        if (it_pad[j])'''
MFMA = '''#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);'''
FULL = 'asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");'
OPS = ': "+v"(hi.x), "+v"(hi.y), "+v"(hi.z), "+v"(hi.w), "+v"(lo.x), "+v"(lo.y), "+v"(lo.z), "+v"(lo.w) : : "memory");'
assert all(k in s for k in (VMOV, STORES, XTOP, SPLIT, MFMA))
b = s.replace(VMOV, INPLACE)
out = eval(expr)
open(f"tools/_r1/hicdiff_amd/csrc/conv_bf16x3_kernel.h", "w").write(out)
PY
    rm -f $R/hicdiff_amd/csrc/conv_bf16x3_ck32.o $R/hicdiff_amd/csrc/conv_bf16x3_ck16.o
    make -C $R/hicdiff_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function ${3:-}" > /dev/null || exit 1
    cp $R/hicdiff_amd/libhicdiff_hip.so $R/lib_r1_$1.so
}
case "${1:-}" in
trees)
    rm -rf $R && mkdir -p $R/hicdiff_amd/csrc $R/include
    for f in $(git ls-tree --name-only 96daf9e hicdiff_amd/csrc/ | grep -v '\.o$'); do git show 96daf9e:$f > $R/$f; done
    for f in include/hicdiff_hip.h include/hicdiff_hip_debug.h; do git show 96daf9e:$f > $R/$f; done
    cp $R/hicdiff_amd/csrc/conv_bf16x3_kernel.h /tmp/r1_orig.h
    variant A 's'
    variant B 'b'
    variant C 'b' '-mllvm -amdgpu-waitcnt-forcezero'
    variant D 'b.replace(INPLACE, INPLACE + "\n            " + FULL)'
    variant E 'b.replace(STORES, STORES[:-7] + "        asm volatile(\"s_nop 1\" " + OPS + "\n    };")'
    variant G 'b.replace(STORES, STORES[:-7] + "        asm volatile(\"s_nop 7\\n\\ts_nop 7\" " + OPS + "\n    };")'
    variant H 'b.replace(XTOP, "        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n" + XTOP)'
    variant I 'b.replace(XTOP, "        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n" + XTOP)'
    variant J 'b.replace(SPLIT, "        " + FULL + "\n" + SPLIT)'
    variant K 'b.replace(MFMA, "            " + FULL + "\n            __builtin_amdgcn_sched_barrier(0);\n" + MFMA)'
    variant L 'b.replace("    conv_epilogue<BM, BN, TM, TN, NT>(", "    " + FULL + "\n    __syncthreads();\n    " + FULL + "\n    conv_epilogue<BM, BN, TM, TN, NT>(")'
    variant M 'b.replace("__syncthreads();", "do { asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_sleep 1\" ::: \"memory\"); __syncthreads(); } while (0);")'
    variant N 'b.replace("__syncthreads();", "do { __syncthreads(); __syncthreads(); } while (0);")'
    ls $R/*.so ;;
run)
    for i in 1 2; do timeout -k 10 600 python tools/ln_zero_lane_probe.py $R/lib_r1_*.so hicdiff_amd/libhicdiff_hip.so 2>&1 | grep -v "amdgpu.ids"; done ;;
*) echo "usage: $0 trees|run"; exit 2 ;;
esac
