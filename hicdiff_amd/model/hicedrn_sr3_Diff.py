"""hicedrn with SR3 noise-level embedding: drop-in for ``src/model/hicedrn_sr3_Diff.py:267-352``."""
from __future__ import annotations

from .hicedrn_Diff import hicedrn_Diff as _Base, n_feat, kernel_size  # noqa: F401


class hicedrn_Diff(_Base):
    _SR3 = True
    _native_train = True       # hd_train_* covers the SR3 flavour too (additive FiLM, continuous noise level)

    def __init__(self, channels=1, out_dim=None, number_resnet=32, self_condition=False,
                 learned_sinusoidal_cond=False, noise_level_emb=True, learned_sinusoidal_dim=16):
        if not noise_level_emb:
            raise NotImplementedError("use hicdiff_amd.model.hicedrn_Diff for the timestep-embedding flavour")
        super().__init__(channels, out_dim, number_resnet, self_condition, learned_sinusoidal_cond, learned_sinusoidal_dim)
        self.noise_level_emb = True
