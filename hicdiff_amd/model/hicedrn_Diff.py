"""hicedrn epsilon-network (32 residual blocks x 256 channels; the net train.py / inference.py
instantiate): drop-in for ``src/model/hicedrn_Diff.py:210-289``."""
from __future__ import annotations

from .. import _lib as L
from .._engine import EpsNetBase, build_param_tree
from .._specs import hicedrn_specs

n_feat = 256
kernel_size = 3


class hicedrn_Diff(EpsNetBase):
    _SR3 = False
    EARLY_BAND_OK = True       # the samplers' precision schedule applies (hicdiff_amd/_diffusion.py:_early_band; profiles/r04_e_*)
    EARLY_BAND_FROM = 0.0      # ... to every step of a long chain: two fp16 products add 2.4-4.0e-4 here (the UNet: 5.5-8.0e-4), profiles/r04_p_*
    _native_train = True       # hd_train_* covers this network (unconditional and self_condition); see hicdiff_amd/_training.py

    def __init__(self, channels=1, out_dim=None, number_resnet=32, self_condition=False,
                 learned_sinusoidal_cond=False, learned_sinusoidal_dim=16):
        super().__init__()
        if learned_sinusoidal_cond:
            raise NotImplementedError("learned sinusoidal embedding is asserted off by GaussianDiffusion (src/hicdiff.py:451)")
        if channels != 1 or (out_dim not in (None, 1)):
            raise NotImplementedError("Hi-C tiles have one channel")
        self.channels = channels
        self.self_condition = self_condition
        self.number_resnet = number_resnet
        self.random_or_learned_sinusoidal_cond = False
        self.out_dim = 1
        build_param_tree(self, hicedrn_specs(channels, number_resnet, self_condition, self._SR3, n_feat))

    def _arch(self) -> L.HdArchDesc:
        a = L.HdArchDesc()
        a.kind, a.dim, a.n_mults = L.HD_ARCH_HICEDRN, n_feat, 0
        a.channels, a.self_condition, a.sr3 = self.channels, int(self.self_condition), int(self._SR3)
        a.groups, a.number_resnet = 1, self.number_resnet
        return a
