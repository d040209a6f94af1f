"""SVD-free degradation operators of the DDRM sampler: drop-in for ``src/functions/svd_replacement.py``.

HiCDiff hard-codes ``deg='deno'`` (inference.py:44, train.py:45), i.e. ``Denoising`` (identity H, all singular values 1),
which ``hd_ddrm_step`` fuses.  The rest of the reference's operator zoo (:72-541) is here too, built from five HIP
primitives -- a column gather (every permutation / selection / zero padding), one small matrix applied to many short
vectors (per-patch and per-pixel factors), the fast Walsh-Hadamard transform, S x S factors on both sides of every image
(the separable blur operators) and a plain dense product (GeneralH) -- no library GEMM.  Index tables and the tiny SVDs
are built once, on the host, at construction.

Conventions are the reference's: vectors go in as (B, ...) and come out as (B, D); the spectral ordering of every
operator (which entry of ``V^T x`` belongs to which singular value) is the reference's, because replayed noise and the
three-case update of ``efficient_generalized_steps`` are defined in that ordering.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib as L


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _flat(vec):
    v = vec.reshape(vec.shape[0], -1)
    if v.dtype != torch.float32:
        v = v.float()
    return v.contiguous()


def gather_cols(src, idx, d_out=None):
    """dst[b][i] = src[b][idx[i]] (0 where idx[i] < 0) on the HIP engine; idx: int32 device tensor."""
    if not src.is_cuda:
        raise RuntimeError("hicdiff_amd runs on MI355X (HIP) tensors only; there is no CPU fallback")
    src = _flat(src)
    d_out = idx.numel() if d_out is None else d_out
    dst = torch.empty((src.shape[0], d_out), device=src.device, dtype=torch.float32)
    rc = L.load().hd_gather_cols(C.c_void_p(src.data_ptr()), C.c_void_p(idx.data_ptr()), C.c_void_p(dst.data_ptr()), src.shape[0], src.shape[1],
                                 d_out, _stream(src.device))
    if rc != 0:
        raise L.HdError(rc, "hd_gather_cols failed")
    return dst


def kvec_matmul(src, mat):
    """mat (K x K) applied to every contiguous K-vector of src (any shape with numel % K == 0)."""
    K = mat.shape[0]
    s = src.contiguous().float()
    dst = torch.empty_like(s)
    rc = L.load().hd_kvec_matmul(C.c_void_p(s.data_ptr()), C.c_void_p(mat.data_ptr()), C.c_void_p(dst.data_ptr()), s.numel() // K, K, _stream(s.device))
    if rc != 0:
        raise L.HdError(rc, "hd_kvec_matmul failed (K <= 64)")
    return dst


def sandwich_matmul(A, x, Bm):
    """A @ x[i] @ Bm for a stack x of S x S images: hd_sandwich_matmul (S <= 64: one workgroup per image with both factors in LDS); larger
    images -- the reference's operators are written for any img_dim -- take two hd_dense_matmul calls, (x Bm) and then A (x Bm) through
    its transpose."""
    S = A.shape[0]
    xs = x.contiguous().float()
    Ac, Bc = A.contiguous().float(), Bm.contiguous().float()        # named: the copies of transposed views must outlive the launch
    if Ac.device != xs.device or Bc.device != xs.device:
        raise ValueError("sandwich_matmul: factors and images must live on the same device")
    n = xs.numel() // (S * S)
    if S > 64:
        t = dense_matmul(xs.reshape(n * S, S), Bc)                                                  # x Bm
        tt = t.reshape(n, S, S).transpose(1, 2).contiguous().reshape(n * S, S)                    # (x Bm)^T
        r = dense_matmul(tt, Ac.t().contiguous())                                                    # (x Bm)^T A^T = (A x Bm)^T
        return r.reshape(n, S, S).transpose(1, 2).contiguous().reshape(xs.shape)
    dst = torch.empty_like(xs)
    rc = L.load().hd_sandwich_matmul(C.c_void_p(Ac.data_ptr()), C.c_void_p(xs.data_ptr()), C.c_void_p(Bc.data_ptr()),
                                     C.c_void_p(dst.data_ptr()), n, S, _stream(xs.device))
    if rc != 0:
        raise L.HdError(rc, "hd_sandwich_matmul failed")
    return dst


def dense_matmul(src, mat):
    """src [N, K] @ mat [K, M]: hd_dense_matmul."""
    s, m = src.contiguous().float(), mat.contiguous().float()
    dst = torch.empty((s.shape[0], m.shape[1]), device=s.device, dtype=torch.float32)
    rc = L.load().hd_dense_matmul(C.c_void_p(s.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(dst.data_ptr()), s.shape[0], s.shape[1], m.shape[1],
                                  _stream(s.device))
    if rc != 0:
        raise L.HdError(rc, "hd_dense_matmul failed")
    return dst


def _idx(values, device):
    return torch.as_tensor(np.asarray(values, dtype=np.int32), device=device)


def _inverse(perm, n):
    inv = np.full(n, -1, dtype=np.int64)
    inv[np.asarray(perm, dtype=np.int64)] = np.arange(len(perm))
    return inv


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
    """Identity operator (:148-168): U = V = I, singular values all one."""

    def __init__(self, channels, img_dim, device):
        self._singulars = torch.ones(channels * img_dim ** 2, device=device)

    @staticmethod
    def _flat(vec):
        return vec.clone().reshape(vec.shape[0], -1)

    V = Vt = U = Ut = add_zeros = lambda self, vec: Denoising._flat(vec)

    def singulars(self):
        return self._singulars


def _pad_zeros(vec, d_total):
    v = _flat(vec)
    out = torch.zeros((v.shape[0], d_total), device=v.device, dtype=torch.float32)
    out[:, :v.shape[1]] = v
    return out


class GeneralH(H_functions):
    """Any dense H through its full SVD (:72-107; memory-hungry upstream too): four dense factors applied by hd_dense_matmul."""

    def __init__(self, H):
        U, s, V = torch.svd(H.detach().float().cpu(), some=False)            # host LAPACK, as the reference on its CPU path
        s = s.clone()
        s[s < 1e-3] = 0
        dev = H.device
        self._U, self._V, self._singulars = U.to(dev), V.to(dev), s.to(dev)
        self._Ut, self._Vt = self._U.T.contiguous(), self._V.T.contiguous()

    def V(self, vec): return dense_matmul(_flat(vec), self._Vt)
    def Vt(self, vec): return dense_matmul(_flat(vec), self._V)
    def U(self, vec): return dense_matmul(_flat(vec), self._Ut)
    def Ut(self, vec): return dense_matmul(_flat(vec), self._U)
    def singulars(self): return self._singulars
    def add_zeros(self, vec): return _pad_zeros(vec, self._V.shape[0])


class Inpainting(H_functions):
    """Keep a subset of the pixels (:110-146): V^T lists the kept pixels first, then the missing ones."""

    def __init__(self, channels, img_dim, missing_indices, device):
        self.channels, self.img_dim = channels, img_dim
        n = channels * img_dim ** 2
        miss = torch.as_tensor(missing_indices).detach().cpu().long().numpy()
        self.missing_indices = torch.as_tensor(miss, device=device)
        kept = np.setdiff1d(np.arange(n), miss)                               # ascending, as the reference's list comprehension
        self.kept_indices = torch.as_tensor(kept, device=device)
        self._singulars = torch.ones(n - miss.shape[0], device=device)
        # pixel order of the reference's spectral vector: (pixel, channel) interleaved -> 'b c p -> b (p c)'
        hw = img_dim ** 2
        pc = (np.arange(n) % channels) * hw + np.arange(n) // channels        # position i of the (p c) vector <- flat (c p) index
        order = np.concatenate([kept, miss])                                  # spectral entry k <- (p c) position order[k]
        self._vt = _idx(pc[order], device)                                    # V^T x [k] = x_flat[pc[order[k]]]
        self._v = _idx(_inverse(pc[order], n), device)                        # its inverse permutation

    def V(self, vec): return gather_cols(vec, self._v)
    def Vt(self, vec): return gather_cols(vec, self._vt)
    def U(self, vec): return _flat(vec).clone()
    def Ut(self, vec): return _flat(vec).clone()
    def singulars(self): return self._singulars
    def add_zeros(self, vec): return _pad_zeros(vec, self.channels * self.img_dim ** 2)


class SuperResolution(H_functions):
    """Block averaging by `ratio` (:171-226): per ratio x ratio patch one singular value; V^T lists every patch's leading
    coefficient first (channel-major), then the remaining ratio^2 - 1 coefficients interleaved patch by patch."""

    def __init__(self, channels, img_dim, ratio, device):
        assert img_dim % ratio == 0
        self.img_dim, self.channels, self.ratio, self.y_dim = img_dim, channels, ratio, img_dim // ratio
        r2, y2 = ratio ** 2, (img_dim // ratio) ** 2
        Hs = torch.full((1, r2), 1.0 / r2)
        U, s, V = torch.svd(Hs, some=False)
        self.U_small, self.singulars_small, self.V_small = U.to(device), s.to(device), V.to(device)
        self.Vt_small = self.V_small.t().contiguous()
        self._V_small_c = self.V_small.contiguous()
        n = channels * img_dim ** 2
        # patchify: element e of patch p of channel c <- pixel (py * r + ey, px * r + ex)
        c, p, e = np.meshgrid(np.arange(channels), np.arange(y2), np.arange(r2), indexing="ij")
        py, px, ey, ex = p // self.y_dim, p % self.y_dim, e // ratio, e % ratio
        patch_src = (c * img_dim + py * ratio + ey) * img_dim + px * ratio + ex              # [c][p][e] -> flat pixel index
        self._patchify = _idx(patch_src.reshape(-1), device)
        self._unpatchify = _idx(_inverse(patch_src.reshape(-1), n), device)
        # reorder: spectral position of coefficient e of patch (c, p): e == 0 -> c * y2 + p; else the strided slots of :205-206
        flat_cp = (c * y2 + p)
        pos = np.where(e == 0, flat_cp, channels * y2 + (e - 1) + flat_cp * (r2 - 1))
        self._to_spec = _idx(_inverse(pos.reshape(-1), n), device)           # spectral[k] <- patches_flat[to_spec[k]]
        self._from_spec = _idx(pos.reshape(-1), device)                      # patches_flat[i] <- spectral[from_spec[i]]

    def V(self, vec):
        patches = gather_cols(vec, self._from_spec)
        return gather_cols(kvec_matmul(patches, self._V_small_c), self._unpatchify)

    def Vt(self, vec):
        patches = gather_cols(vec, self._patchify)
        return gather_cols(kvec_matmul(patches, self.Vt_small), self._to_spec)

    def U(self, vec): return self.U_small[0, 0] * _flat(vec)
    def Ut(self, vec): return self.U_small[0, 0] * _flat(vec)
    def singulars(self): return self.singulars_small.repeat(self.channels * self.y_dim ** 2)
    def add_zeros(self, vec): return _pad_zeros(vec, _flat(vec).shape[1] * self.ratio ** 2)


class Colorization(H_functions):
    """Grey value from three channels (:229-272): one 3-vector ("needle") per pixel."""

    def __init__(self, img_dim, device):
        self.channels, self.img_dim = 3, img_dim
        U, s, V = torch.svd(torch.tensor([[0.3333, 0.3334, 0.3333]]), some=False)
        self.U_small, self.singulars_small, self.V_small = U.to(device), s.to(device), V.to(device)
        self.Vt_small = self.V_small.t().contiguous()
        self._V_small_c = self.V_small.contiguous()
        hw = img_dim ** 2
        pc = np.arange(3 * hw)
        self._to_needles = _idx((pc % 3) * hw + pc // 3, device)             # (p c) position <- flat (c p) index
        self._from_needles = _idx(_inverse((pc % 3) * hw + pc // 3, 3 * hw), device)

    def V(self, vec):
        return gather_cols(kvec_matmul(gather_cols(vec, self._to_needles), self._V_small_c), self._from_needles)

    def Vt(self, vec):
        return gather_cols(kvec_matmul(gather_cols(vec, self._to_needles), self.Vt_small), self._from_needles)

    def U(self, vec): return self.U_small[0, 0] * _flat(vec)
    def Ut(self, vec): return self.U_small[0, 0] * _flat(vec)
    def singulars(self): return self.singulars_small.repeat(self.img_dim ** 2)
    def add_zeros(self, vec): return _pad_zeros(vec, self.channels * self.img_dim ** 2)


class WalshHadamardCS(H_functions):
    """Compressive sensing with a permuted Walsh-Hadamard basis (:275-318): V = FWHT o scatter(perm)."""

    def __init__(self, channels, img_dim, ratio, perm, device):
        self.channels, self.img_dim, self.ratio = channels, img_dim, ratio
        hw = img_dim ** 2
        if hw & (hw - 1):
            raise ValueError("WalshHadamardCS needs img_dim^2 to be a power of two")
        self.perm = torch.as_tensor(perm).to(device)
        pm = self.perm.detach().cpu().long().numpy()
        self._singulars = torch.ones(channels * hw // ratio, device=device)
        n = channels * hw
        # V: temp[c][perm[p]] = vec[(p c)][p * C + c]  ->  temp_flat[c * hw + q] <- vec[inv_perm[q] * C + c]
        inv = _inverse(pm, hw)
        cc, q = np.meshgrid(np.arange(channels), np.arange(hw), indexing="ij")
        self._v_in = _idx((inv[q] * channels + cc).reshape(-1), device)
        # V^T: out[(p c)] = fwht(vec)[c][perm[p]]
        p, c2 = np.meshgrid(np.arange(hw), np.arange(channels), indexing="ij")
        self._vt_out = _idx((c2 * hw + pm[p]).reshape(-1), device)
        self._n = n

    def fwht(self, vec):
        a = _flat(vec).clone()
        hw = self.img_dim ** 2
        rc = L.load().hd_fwht(C.c_void_p(a.data_ptr()), a.shape[0] * self.channels, hw, 1.0 / self.img_dim, _stream(a.device))
        if rc != 0:
            raise L.HdError(rc, "hd_fwht failed")
        return a.reshape(vec.shape[0], self.channels, hw)

    def V(self, vec): return self.fwht(gather_cols(vec, self._v_in)).reshape(vec.shape[0], -1)
    def Vt(self, vec): return gather_cols(self.fwht(vec), self._vt_out)
    def U(self, vec): return _flat(vec).clone()
    def Ut(self, vec): return _flat(vec).clone()
    def singulars(self): return self._singulars
    def add_zeros(self, vec): return _pad_zeros(vec, self._n)


def _conv_matrix(kernel, rows, img_dim, stride=1, reflect=False):
    """The 1-D convolution matrices of :338-349 (strided, reflective padding) and :415-421 (same size, zero padding)."""
    k = torch.as_tensor(kernel).detach().float().cpu()
    K = k.shape[0]
    Hs = torch.zeros(rows, img_dim)
    if reflect:
        for i in range(stride // 2, img_dim + stride // 2, stride):
            for j in range(i - K // 2, i + K // 2):
                je = j
                if je < 0:
                    je = -je - 1
                if je >= img_dim:
                    je = (img_dim - 1) - (je - img_dim)
                Hs[i // stride, je] += k[j - i + K // 2]
    else:
        for i in range(img_dim):
            for j in range(i - K // 2, i + K // 2):
                if 0 <= j < img_dim:
                    Hs[i, j] = k[j - i + K // 2]
    return Hs


class _Separable(H_functions):
    """Shared body of the operators whose H is a Kronecker product of two small matrices acting on the rows and columns of
    every channel image: V x = V1 X V2^T etc. are S x S products on both sides of every image (hd_sandwich_matmul); the singular-value ordering is a column gather."""

    def _lr(self, A, vec, Bm, dim):
        x = _flat(vec).reshape(vec.shape[0] * self.channels, dim, dim)
        return sandwich_matmul(A, x, Bm).reshape(vec.shape[0], self.channels, -1)


class Deblurring(_Separable):
    """Separable blur (:401-470): singular values = outer product of the 1-D ones, sorted descending."""

    def __init__(self, kernel, channels, img_dim, device, ZERO=3e-2, _svd=None):
        self.img_dim, self.channels = img_dim, channels
        U, s, V = _svd if _svd is not None else torch.svd(_conv_matrix(kernel, img_dim, img_dim), some=False)
        s = s.clone().float().cpu()
        s[s < ZERO] = 0
        big = torch.matmul(s.reshape(img_dim, 1), s.reshape(1, img_dim)).reshape(img_dim ** 2)
        self._singulars, perm = big.sort(descending=True)
        self._finish(U, V, U, V, perm, device)

    def _finish(self, U1, V1, U2, V2, perm, device):
        d = device
        self.U_small1, self.V_small1, self.U_small2, self.V_small2 = U1.float().to(d), V1.float().to(d), U2.float().to(d), V2.float().to(d)
        self.U_small, self.V_small = self.U_small1, self.V_small1
        self._singulars = self._singulars.to(d)
        self._perm = perm.to(d)
        hw, Cn = self.img_dim ** 2, self.channels
        pm = perm.cpu().long().numpy()
        k, c = np.meshgrid(np.arange(hw), np.arange(Cn), indexing="ij")                 # spectral (k c) position
        self._to_spec = _idx((c * hw + pm[k]).reshape(-1), device)                       # spec[(k c)] <- img[c][perm[k]]
        self._from_spec = _idx(_inverse((c * hw + pm[k]).reshape(-1), Cn * hw), device)  # img[c][q] <- spec[...]

    def V(self, vec):
        img = gather_cols(vec, self._from_spec)
        return self._lr(self.V_small1, img, self.V_small2.t(), self.img_dim).reshape(vec.shape[0], -1)

    def Vt(self, vec):
        return gather_cols(self._lr(self.V_small1.t(), vec, self.V_small2, self.img_dim), self._to_spec)

    def U(self, vec):
        img = gather_cols(vec, self._from_spec)
        return self._lr(self.U_small1, img, self.U_small2.t(), self.img_dim).reshape(vec.shape[0], -1)

    def Ut(self, vec):
        return gather_cols(self._lr(self.U_small1.t(), vec, self.U_small2, self.img_dim), self._to_spec)

    def singulars(self): return self._singulars.repeat(1, self.channels).reshape(-1)
    def add_zeros(self, vec): return _flat(vec).clone()


class Deblurring2D(Deblurring):
    """Anisotropic blur (:473-541): different 1-D kernels along rows and columns."""

    def __init__(self, kernel1, kernel2, channels, img_dim, device, _svd=None):
        self.img_dim, self.channels = img_dim, channels
        if _svd is None:
            U1, s1, V1 = torch.svd(_conv_matrix(kernel1, img_dim, img_dim), some=False)
            U2, s2, V2 = torch.svd(_conv_matrix(kernel2, img_dim, img_dim), some=False)
        else:
            (U1, s1, V1), (U2, s2, V2) = _svd
        s1, s2 = s1.clone().float().cpu(), s2.clone().float().cpu()
        s1[s1 < 3e-2] = 0
        s2[s2 < 3e-2] = 0
        big = torch.matmul(s1.reshape(img_dim, 1), s2.reshape(1, img_dim)).reshape(img_dim ** 2)
        self._singulars, perm = big.sort(descending=True)
        self._finish(U1, V1, U2, V2, perm, device)


class SRConv(_Separable):
    """Convolutional (e.g. bicubic) down-sampling (:321-398): strided 1-D convolution with reflective padding along both axes."""

    def __init__(self, kernel, channels, img_dim, device, stride=1, _svd=None):
        self.img_dim, self.channels, self.ratio = img_dim, channels, stride
        sd = self.small_dim = img_dim // stride
        U, s, V = _svd if _svd is not None else torch.svd(_conv_matrix(kernel, sd, img_dim, stride, reflect=True), some=False)
        s = s.clone().float().cpu()
        s[s < 3e-2] = 0
        self.U_small, self.V_small, self.singulars_small = U.float().to(device), V.float().to(device), s.to(device)
        self._singulars = torch.matmul(s.reshape(sd, 1), s.reshape(1, sd)).reshape(sd ** 2).to(device)
        pm = np.asarray([img_dim * i + j for i in range(sd) for j in range(sd)] + [img_dim * i + j for i in range(sd) for j in range(sd, img_dim)])
        self._perm = torch.as_tensor(pm, device=device)
        hw, Cn, npm = img_dim ** 2, channels, len(pm)
        # V^T output (k c): k < len(perm): image entry perm[k]; k >= len(perm): image entry k (left in place, :373)
        src_k = np.concatenate([pm, np.arange(npm, hw)])
        k, c = np.meshgrid(np.arange(hw), np.arange(Cn), indexing="ij")
        self._to_spec = _idx((c * hw + src_k[k]).reshape(-1), device)
        # V input: temp[perm[k]] = vec[k] for k < len(perm); temp[q] = vec[q] for q >= len(perm) (applied second, :359-360)
        img_src = np.full(hw, -1, dtype=np.int64)
        img_src[pm] = np.arange(npm)
        img_src[npm:] = np.arange(npm, hw)
        c2, q = np.meshgrid(np.arange(Cn), np.arange(hw), indexing="ij")
        self._from_spec = _idx((img_src[q] * Cn + c2).reshape(-1), device)
        k2, c3 = np.meshgrid(np.arange(sd ** 2), np.arange(Cn), indexing="ij")
        self._u_in = _idx(_inverse((c3 * sd ** 2 + k2).reshape(-1), Cn * sd ** 2), device)   # (k c) -> (c k)
        self._u_out = _idx((c3 * sd ** 2 + k2).reshape(-1), device)                            # (c k) -> (k c)

    def V(self, vec):
        img = gather_cols(vec, self._from_spec)
        return self._lr(self.V_small, img, self.V_small.t(), self.img_dim).reshape(vec.shape[0], -1)

    def Vt(self, vec):
        return gather_cols(self._lr(self.V_small.t(), vec, self.V_small, self.img_dim), self._to_spec)

    def U(self, vec):
        img = gather_cols(vec, self._u_in)
        return self._lr(self.U_small, img, self.U_small.t(), self.small_dim).reshape(vec.shape[0], -1)

    def Ut(self, vec):
        return gather_cols(self._lr(self.U_small.t(), vec, self.U_small, self.small_dim), self._u_out)

    def singulars(self): return self._singulars.repeat_interleave(3).reshape(-1)      # as upstream: written for 3 channels (:390)
    def add_zeros(self, vec): return _pad_zeros(vec, _flat(vec).shape[1] * self.ratio ** 2)
