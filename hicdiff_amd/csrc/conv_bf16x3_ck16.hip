// bf16x3 convolution, K slice of 16 channels (layers whose channel count is not a multiple of 32):
// instantiations and dispatch (kernel: conv_bf16x3_kernel.h).
#include "conv_bf16x3_kernel.h"

// kernel pointer + its profiler name, spelled as rocprofv3 prints the instantiation (built only while profiling)
#define K16(WM, WN, TM, TN, MAXI, NTAPS)                                                                   \
    conv_igemm_bf16x3_kernel<WM, WN, TM, TN, 16, MAXI, MODE, NTAPS>,                                              \
        conv_prof_name("conv_igemm_bf16x3_kernel<" #WM ", " #WN ", " #TM ", " #TN ", 16, " #MAXI ", ", MODE, ", " #NTAPS ">")

#define KF16(WM, WN, TM, TN, MAXI)                                                                          \
    conv_igemm_f16w2_kernel<WM, WN, TM, TN, 16, MAXI, MODE, 9>,                                                    \
        conv_prof_name("conv_igemm_f16w2_kernel<" #WM ", " #WN ", " #TM ", " #TN ", 16, " #MAXI ", ", MODE, ", 9>")

#define KF161(WM, WN, TM, TN, MAXI)                                                                         \
    conv_igemm_f16w1_kernel<WM, WN, TM, TN, 16, MAXI, MODE, 9>,                                                    \
        conv_prof_name("conv_igemm_f16w1_kernel<" #WM ", " #WN ", " #TM ", " #TN ", 16, " #MAXI ", ", MODE, ", 9>")

template <int MODE>
static int launch_mode(ConvLaunch& L, hipStream_t st) {
    const bool t9 = L.k.KH == 3 && L.k.KW == 3 && MODE != IN_LAYERNORM;
    if constexpr (MODE != IN_LAYERNORM) {
        if constexpr (MODE != IN_AFFINE_SILU_E) if (t9 && L.k.f16w2 == 2) {
            switch (L.cfg) {
                case 0: return launch_one(KF161(2, 2, 2, 2, 3), L, st);
                case 1: return launch_one(KF161(2, 2, 2, 1, 3), L, st);
                case 2: return launch_one(KF161(4, 1, 2, 2, 3), L, st);
                default: return launch_one(KF161(4, 2, 2, 2, 2), L, st, 512);
            }
        }
        if constexpr (MODE != IN_AFFINE_SILU_E) if (t9 && L.k.f16w2) {
            switch (L.cfg) {
                case 0: return launch_one(KF16(2, 2, 2, 2, 3), L, st);
                case 1: return launch_one(KF16(2, 2, 2, 1, 3), L, st);
                case 2: return launch_one(KF16(4, 1, 2, 2, 3), L, st);
                default: return launch_one(KF16(4, 2, 2, 2, 2), L, st, 512);
            }
        }
        if (t9) {
            switch (L.cfg) {
                case 0: return launch_one(K16(2, 2, 2, 2, 3, 9), L, st);
                case 1: return launch_one(K16(2, 2, 2, 1, 3, 9), L, st);
                case 2: return launch_one(K16(4, 1, 2, 2, 3, 9), L, st);
                default: return launch_one(K16(4, 2, 2, 2, 2, 9), L, st, 512);
            }
        }
    }
    switch (L.cfg) {
        case 0: return launch_one(K16(2, 2, 2, 2, 4, 0), L, st);
        case 1: return launch_one(K16(2, 2, 2, 1, 4, 0), L, st);
        case 2: return launch_one(K16(4, 1, 2, 2, 4, 0), L, st);
    }
    hd_set_error("conv: no bf16x3 kernel variant for this tile"); return -1;
}

int launch_conv_bf16x3_ck16(ConvLaunch& L, hipStream_t st) {
    switch (conv_kernel_mode(L)) {
        case IN_AFFINE_SILU: return launch_mode<IN_AFFINE_SILU>(L, st);
        case IN_AFFINE_SILU_E: return launch_mode<IN_AFFINE_SILU_E>(L, st);
        case IN_LAYERNORM: return launch_mode<IN_LAYERNORM>(L, st);
        case IN_SOFTMAX32: hd_set_error("conv: the softmax loader needs 32-channel K slices"); return -1;
        default: return launch_mode<IN_NONE>(L, st);
    }
}

int conv_bf16x3_max_items(int cfg, int ck, bool taps9, bool layernorm) {
    const bool t9 = taps9 && !layernorm;
    if (ck == 32) {
        if (cfg == 3) return 3;
        if (cfg == 2) return t9 ? 6 : 4;
        return t9 ? 5 : 8;
    }
    if (cfg == 3) return 2;
    return t9 ? 3 : 4;
}

int launch_conv_bf16x3_ck32(ConvLaunch& L, hipStream_t st);
int launch_conv_bf16x3(ConvLaunch& L, hipStream_t st) {
    return L.ck == 32 ? launch_conv_bf16x3_ck32(L, st) : launch_conv_bf16x3_ck16(L, st);
}
