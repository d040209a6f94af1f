// Plain-bf16 forms (ONE bf16 MFMA per product) of the 4-wave convolution kernels, K slice of 32 channels: the optional bf16 arithmetic of the
// native training step (train.hip / train_unet.inc, `plain_bf16`) on every tile shape the UNet's layers take.  The 8-wave 256 x 128 form lives
// in conv_bf16x3_ck32.hip; before round 4 it was the only one, so the UNet's bf16 step still ran most of its convolutions on three products.
// Same body, AR = 1 (conv_bf16x3_kernel.h): no lo half of the window or of the weight fragments is written or read.
#include "conv_bf16x3_kernel.h"

template <int WM, int WN, int TM, int TN, int CK, int MAXI, int MODE, int NTAPS>
__global__ __launch_bounds__(64 * WM * WN, (NTAPS == 0 && MAXI > 4) ? 1 : 2) void conv_igemm_bf16_4w_kernel(ConvKArgs p) {
    conv_igemm_bf16x3_body<WM, WN, TM, TN, CK, MAXI, MODE, NTAPS, 1>(p);
}

#define KP32(WM, WN, TM, TN, MAXI, NTAPS)                                                                  \
    conv_igemm_bf16_4w_kernel<WM, WN, TM, TN, 32, MAXI, MODE, NTAPS>,                                             \
        conv_prof_name("conv_igemm_bf16_kernel<" #WM ", " #WN ", " #TM ", " #TN ", 32, " #MAXI ", ", MODE, ", " #NTAPS ">")

template <int MODE>
static int launch_plain_mode(ConvLaunch& L, hipStream_t st) {
    const bool t9 = L.k.KH == 3 && L.k.KW == 3;
    const int need = (L.k.npx * 4 + 255) / 256;
    if (t9) {
        switch (L.cfg) {
            case 0: return need <= 3 ? launch_one(KP32(2, 2, 2, 2, 3, 9), L, st) : launch_one(KP32(2, 2, 2, 2, 5, 9), L, st);
            case 1: return need <= 3 ? launch_one(KP32(2, 2, 2, 1, 3, 9), L, st) : launch_one(KP32(2, 2, 2, 1, 5, 9), L, st);
            case 2: return launch_one(KP32(4, 1, 2, 2, 6, 9), L, st);
        }
        return 1;
    }
    switch (L.cfg) {
        case 0: return need <= 2 ? launch_one(KP32(2, 2, 2, 2, 2, 0), L, st) : launch_one(KP32(2, 2, 2, 2, 8, 0), L, st);
        case 1: return need <= 2 ? launch_one(KP32(2, 2, 2, 1, 2, 0), L, st) : launch_one(KP32(2, 2, 2, 1, 8, 0), L, st);
        case 2: return launch_one(KP32(4, 1, 2, 2, 4, 0), L, st);
    }
    return 1;
}

// 1: no plain form for this launch (the caller takes the three-product kernel); otherwise launch_one's result
int launch_conv_bf16_plain_ck32(ConvLaunch& L, hipStream_t st) {
    static const bool off = getenv("HICDIFF_PLAIN_4W") && atoi(getenv("HICDIFF_PLAIN_4W")) == 0;      // A/B switch: the round-3 state (8-wave form only)
    if (off || !L.k.plain || L.cfg > 2 || L.k.f16w2) return 1;
    switch (conv_kernel_mode(L)) {
        case IN_NONE: return launch_plain_mode<IN_NONE>(L, st);
        case IN_AFFINE_SILU: return launch_plain_mode<IN_AFFINE_SILU>(L, st);
        case IN_AFFINE_SILU_E: return launch_plain_mode<IN_AFFINE_SILU_E>(L, st);
        default: return 1;
    }
}
