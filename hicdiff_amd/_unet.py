"""Unet epsilon-network handle shared by the three diffusion flavours (reference constructors:
src/hicdiff.py:256-269, src/hicdiff_condition.py:256-269, src/hicdiff_sr3.py:311-324)."""
from __future__ import annotations

from . import _lib as L
from ._engine import EpsNetBase, build_param_tree
from ._specs import unet_specs


class UnetBase(EpsNetBase):
    EARLY_BAND_FROM = 0.5     # the early band of the precision schedule is t >= T / 2
    EARLY_BAND_OK = True      # the samplers' precision schedule applies (hicdiff_amd/_diffusion.py:_early_band; measured: profiles/r04_e_*)

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=1, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16, noise_level_emb=False):
        super().__init__()
        if learned_variance or learned_sinusoidal_cond or random_fourier_features:
            # GaussianDiffusion asserts these off (src/hicdiff.py:450-451); no reference driver enables them
            raise NotImplementedError("learned_variance / learned or random sinusoidal embeddings are not part of the HiCDiff path")
        if init_dim not in (None, dim):
            raise NotImplementedError("init_dim != dim is not used by the reference drivers")
        if channels != 1:
            raise NotImplementedError("Hi-C tiles have one channel")
        if dim % 16 or dim % resnet_block_groups:
            raise ValueError("dim must be a multiple of 16 and of resnet_block_groups")
        self.dim = dim
        self.dim_mults = tuple(dim_mults)
        self.channels = channels
        self.self_condition = self_condition
        self.resnet_block_groups = resnet_block_groups
        self.noise_level_emb = noise_level_emb
        # native training step (csrc/train_unet.inc): width a multiple of 64, at most four levels
        self._native_train = dim % 64 == 0 and len(self.dim_mults) <= 4
        self.random_or_learned_sinusoidal_cond = False
        self.out_dim = out_dim if out_dim is not None else channels
        if self.out_dim != 1:
            raise NotImplementedError("out_dim must be 1")
        build_param_tree(self, unet_specs(dim, self.dim_mults, channels, self_condition, noise_level_emb))

    def _arch(self) -> L.HdArchDesc:
        a = L.HdArchDesc()
        a.kind, a.dim, a.n_mults = L.HD_ARCH_UNET, self.dim, len(self.dim_mults)
        for i, m in enumerate(self.dim_mults):
            a.mults[i] = m
        a.channels, a.self_condition, a.sr3 = self.channels, int(self.self_condition), int(self.noise_level_emb)
        a.groups, a.number_resnet = self.resnet_block_groups, 0
        return a
