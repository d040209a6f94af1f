"""SVD-free degradation operators for the DDRM sampler; only what HiCDiff selects.

The reference vendors DDRM's whole operator zoo (src/functions/svd_replacement.py:72-541) but
hard-codes ``deg='deno'`` (inference.py:44, train.py:45), i.e. ``Denoising`` (:148-168): H = I, all
singular values 1.  The other degradations are out of scope.
"""
from __future__ import annotations

import torch


class H_functions:
    """Interface of src/functions/svd_replacement.py:3-69: vectors are (B, ...) in, (B, D) out."""

    def V(self, vec): raise NotImplementedError()
    def Vt(self, vec): raise NotImplementedError()
    def U(self, vec): raise NotImplementedError()
    def Ut(self, vec): raise NotImplementedError()
    def singulars(self): raise NotImplementedError()
    def add_zeros(self, vec): raise NotImplementedError()

    def H(self, vec):
        s = self.singulars()
        return self.U(s * self.Vt(vec)[:, :s.shape[0]])

    def Ht(self, vec):
        s = self.singulars()
        return self.V(self.add_zeros(s * self.Ut(vec)[:, :s.shape[0]]))

    def H_pinv(self, vec):
        s = self.singulars()
        tmp = self.Ut(vec)
        tmp[:, :s.shape[0]] = tmp[:, :s.shape[0]] / s
        return self.V(self.add_zeros(tmp))


class Denoising(H_functions):
    """Identity operator: U = V = I, singular values all one."""

    def __init__(self, channels, img_dim, device):
        self._singulars = torch.ones(channels * img_dim ** 2, device=device)

    @staticmethod
    def _flat(vec):
        return vec.clone().reshape(vec.shape[0], -1)

    V = Vt = U = Ut = add_zeros = lambda self, vec: Denoising._flat(vec)

    def singulars(self):
        return self._singulars
