// bf16x3 convolution, K slice of 32 channels: instantiations and dispatch (kernel: conv_bf16x3_kernel.h).
#include "conv_bf16x3_kernel.h"

// kernel pointer + its profiler name, spelled as rocprofv3 prints the instantiation (built only while profiling)
#define K32(WM, WN, TM, TN, MAXI, NTAPS)                                                                   \
    conv_igemm_bf16x3_kernel<WM, WN, TM, TN, 32, MAXI, MODE, NTAPS>,                                              \
        conv_prof_name("conv_igemm_bf16x3_kernel<" #WM ", " #WN ", " #TM ", " #TN ", 32, " #MAXI ", ", MODE, ", " #NTAPS ">")

#define KF32(WM, WN, TM, TN, MAXI)                                                                          \
    conv_igemm_f16w2_kernel<WM, WN, TM, TN, 32, MAXI, MODE, 9>,                                                    \
        conv_prof_name("conv_igemm_f16w2_kernel<" #WM ", " #WN ", " #TM ", " #TN ", 32, " #MAXI ", ", MODE, ", 9>")

#define KF321(WM, WN, TM, TN, MAXI)                                                                         \
    conv_igemm_f16w1_kernel<WM, WN, TM, TN, 32, MAXI, MODE, 9>,                                                    \
        conv_prof_name("conv_igemm_f16w1_kernel<" #WM ", " #WN ", " #TM ", " #TN ", 32, " #MAXI ", ", MODE, ", 9>")

template <int MODE>
static int launch_mode(ConvLaunch& L, hipStream_t st) {
    const bool t9 = L.k.KH == 3 && L.k.KW == 3 && MODE != IN_LAYERNORM && MODE != IN_SOFTMAX32;
    const int nt = L.cfg == 3 ? 512 : 256;
    const int need = (L.k.npx * 4 + nt - 1) / nt;
    if constexpr (MODE != IN_LAYERNORM && MODE != IN_SOFTMAX32) {
        if constexpr (MODE != IN_AFFINE_SILU_E) if (t9 && L.k.f16w2 == 2) {       // one fp16 product per multiply
            switch (L.cfg) {
                case 0: return need <= 3 ? launch_one(KF321(2, 2, 2, 2, 3), L, st) : launch_one(KF321(2, 2, 2, 2, 5), L, st);
                case 1: return need <= 3 ? launch_one(KF321(2, 2, 2, 1, 3), L, st) : launch_one(KF321(2, 2, 2, 1, 5), L, st);
                case 2: return launch_one(KF321(4, 1, 2, 2, 6), L, st);
                default: return launch_one(KF321(4, 2, 2, 2, 3), L, st, 512);
            }
        }
        if constexpr (MODE != IN_AFFINE_SILU_E) if (t9 && L.k.f16w2) {       // two fp16 products per multiply (conv_host.hip decides; the weight pointer is the fp16 image then)
            switch (L.cfg) {
                case 0: return need <= 3 ? launch_one(KF32(2, 2, 2, 2, 3), L, st) : launch_one(KF32(2, 2, 2, 2, 5), L, st);
                case 1: return need <= 3 ? launch_one(KF32(2, 2, 2, 1, 3), L, st) : launch_one(KF32(2, 2, 2, 1, 5), L, st);
                case 2: return launch_one(KF32(4, 1, 2, 2, 6), L, st);
                default: return launch_one(KF32(4, 2, 2, 2, 3), L, st, 512);
            }
        }
        if (t9) {
            switch (L.cfg) {
                case 0: return need <= 3 ? launch_one(K32(2, 2, 2, 2, 3, 9), L, st) : launch_one(K32(2, 2, 2, 2, 5, 9), L, st);
                case 1: return need <= 3 ? launch_one(K32(2, 2, 2, 1, 3, 9), L, st) : launch_one(K32(2, 2, 2, 1, 5, 9), L, st);
                case 2: return launch_one(K32(4, 1, 2, 2, 6, 9), L, st);
                default:
                    if (MODE == IN_NONE && (L.k.ep & EP_FILM_SILU_BWD)) {        // training: the data gradient through FiLM + SiLU (its own instantiations)
                        if (L.k.plain) return launch_one(conv_igemm_bf16x3_fbwd_kernel<4, 2, 2, 2, 32, 3, IN_NONE, 9, true>, hd_prof_is_on() ? "conv_igemm_bf16_kernel<4, 2, 2, 2, 32, 3, 0, 9> (FiLM bwd)" : nullptr, L, st, 512);
                        return launch_one(conv_igemm_bf16x3_fbwd_kernel<4, 2, 2, 2, 32, 3, IN_NONE, 9, false>, hd_prof_is_on() ? "conv_igemm_bf16x3_kernel<4, 2, 2, 2, 32, 3, 0, 9> (FiLM bwd)" : nullptr, L, st, 512);
                    }
                    if (MODE == IN_NONE && L.k.plain)
                        return launch_one(conv_igemm_bf16_kernel<4, 2, 2, 2, 32, 3, IN_NONE, 9>, hd_prof_is_on() ? "conv_igemm_bf16_kernel<4, 2, 2, 2, 32, 3, 0, 9>" : nullptr, L, st, 512);
                    return launch_one(K32(4, 2, 2, 2, 3, 9), L, st, 512);
            }
        }
    }
    switch (L.cfg) {
        case 0: return need <= 2 ? launch_one(K32(2, 2, 2, 2, 2, 0), L, st) : launch_one(K32(2, 2, 2, 2, 8, 0), L, st);
        case 1: return need <= 2 ? launch_one(K32(2, 2, 2, 1, 2, 0), L, st) : launch_one(K32(2, 2, 2, 1, 8, 0), L, st);
        case 2: return launch_one(K32(4, 1, 2, 2, 4, 0), L, st);
    }
    hd_set_error("conv: no bf16x3 kernel variant for this tile"); return -1;
}

int launch_conv_bf16_plain_ck32(ConvLaunch& L, hipStream_t st);   // conv_bf16_plain_ck32.hip: 1 = no plain-bf16 form of this launch

int launch_conv_bf16x3_ck32(ConvLaunch& L, hipStream_t st) {
    if (L.k.plain && L.cfg <= 2) {
        const int rc = launch_conv_bf16_plain_ck32(L, st);
        if (rc != 1) return rc;
    }
    switch (conv_kernel_mode(L)) {
        case IN_AFFINE_SILU: return launch_mode<IN_AFFINE_SILU>(L, st);
        case IN_AFFINE_SILU_E: return launch_mode<IN_AFFINE_SILU_E>(L, st);
        case IN_LAYERNORM: return launch_mode<IN_LAYERNORM>(L, st);
        case IN_SOFTMAX32: return launch_mode<IN_SOFTMAX32>(L, st);
        default: return launch_mode<IN_NONE>(L, st);
    }
}
