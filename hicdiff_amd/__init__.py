"""hicdiff_amd -- MI355X-native (gfx950) implementation of HiCDiff's DDPM/DDRM sampling hot path.

Module names mirror the reference package so that ``from src.hicdiff import Unet, GaussianDiffusion``
becomes ``from hicdiff_amd.hicdiff import Unet, GaussianDiffusion``:

    hicdiff_amd.hicdiff            <- src/hicdiff.py            (unconditional)
    hicdiff_amd.hicdiff_condition  <- src/hicdiff_condition.py  (conditioned on the low-coverage tile)
    hicdiff_amd.hicdiff_sr3        <- src/hicdiff_sr3.py        (SR3 noise-level conditioning)
    hicdiff_amd.model.hicedrn_Diff / hicedrn_sr3_Diff           <- src/model/...
    hicdiff_amd.functions.denoising / svd_replacement / H_func  <- src/functions/... (DDRM, 'deno')

The compute lives in libhicdiff_hip.so (hand-written HIP, C ABI in include/hicdiff_hip.h); these
modules are the host-side mirror of the reference's object contract.  There is no CPU fallback.
"""
__version__ = "0.1.0"
