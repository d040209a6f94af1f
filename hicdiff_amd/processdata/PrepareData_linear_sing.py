"""Tile producer / stitcher on the HIP engine behind the reference's names.

  splitPieces(fn, piece_size, step, resol)   processdata/PrepareData_linear_sing.py:25-46 -- same arguments, same
                                             (n,1,piece,piece) float array back; the cut itself is ``hd_split_pieces``
  tile_origins / split_pieces_device         the same on device tensors (no file, no host copy)
  stitch_pieces_device / stitchPieces        the inverse the reference lacks: tiles -> dense symmetric matrix
                                             (``hd_stitch_pieces``)
  GSE130711Module / GSE131811Module          the DataModule contract of :106-343 / :345-593: directory and file names,
                                             ``split_numpy`` from ``Full_Mats/``, the chromosome splits, and the
                                             (noisy, target, sample, chromosome) items of ``gse131811Dataset``

Not here (out of scope): reading ``.mcool`` files with cooler and the text-file detour of
``extract_constraint_mats`` / ``loadBothConstraints`` (:48-103,129-181) -- ``Full_Mats/*.npy`` is where this build
picks the pipeline up, and both methods say so when called.
"""
from __future__ import annotations

import ctypes as C
import glob
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .. import _lib as L
from ..functions.H_func import MakeFunc


def tile_origins(n, piece_size, step, resol):
    """(row, col) of every tile splitPieces cuts from an n x n matrix, in its output order, and the padded size.

    The reference's loop (:39-43): pad n up to a multiple of piece_size; i over range(0, bound, step), j over
    range(i, bound, step); keep |i-j| <= int(piece_size*4*scal+1) with scal = int(40000/resol), and i+step, j+step <= bound.
    """
    if piece_size < 1 or step < 1:
        raise ValueError("piece_size and step must be positive")
    scal = int(40000 / resol)
    rest = n % piece_size
    bound = n if rest == 0 else n + piece_size - rest
    band = int(piece_size * 4 * scal + 1)
    starts = np.arange(0, bound, step, dtype=np.int64)
    starts = starts[starts + step <= bound]
    if len(starts) and starts[-1] + piece_size > bound:
        # step < piece_size: the reference's slices run past the padded edge and np.asarray fails on the ragged list
        raise ValueError("setting an array element with a sequence: the last tiles run past the padded matrix (step < piece_size)")
    ii, jj = np.meshgrid(starts, starts, indexing="ij")
    keep = (jj >= ii) & (jj - ii <= band)
    return np.stack([ii[keep], jj[keep]], axis=1).astype(np.int64), int(bound)     # row-major = the loop order


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def split_pieces_device(mat, piece_size, step, resol):
    """mat: (n,n) float32 ROCm tensor -> ((ntiles,1,piece,piece) tensor, origins int64 ndarray)."""
    if not mat.is_cuda:
        raise RuntimeError("hicdiff_amd.processdata runs on the GPU only: move the matrix to a ROCm device")
    if mat.dim() != 2 or mat.shape[0] != mat.shape[1]:
        raise AssertionError("contact matrix must be square")                      # `assert bound == bound1` (:30)
    lib = L.load()
    m = mat.detach().to(torch.float32).contiguous()
    n = m.shape[0]
    org, _ = tile_origins(n, piece_size, step, resol)
    tiles = torch.empty((len(org), 1, piece_size, piece_size), dtype=torch.float32, device=m.device)
    if len(org):
        o = torch.from_numpy(org.astype(np.int32)).to(m.device)
        rc = lib.hd_split_pieces(_ptr(m), n, _ptr(o), len(org), piece_size, _ptr(tiles), _stream(m))
        if rc != 0:
            raise L.HdError(rc, "hd_split_pieces failed")
    return tiles, org


def stitch_table(origins, n, step):
    """int32 [nb][nb] lookup over the step grid for hd_stitch_pieces: tile index at (row//step, col//step) or -1."""
    org = np.asarray(origins, dtype=np.int64).reshape(-1, 2)
    nb = max(1, -(-max(n, int(org.max()) + 1 if len(org) else 1) // step))
    table = np.full((nb, nb), -1, dtype=np.int32)
    if len(org):
        table[org[:, 0] // step, org[:, 1] // step] = np.arange(len(org), dtype=np.int32)
    return table


def stitch_pieces_device(tiles, origins, n, step=None):
    """tiles: (ntiles,1,p,p) or (ntiles,p,p) ROCm tensor cut at `origins` -> dense (n,n) matrix.

    Element (r,c) takes the tile element that holds it, else the one that holds (c,r), else 0; only upper-triangle
    origins (row <= col) on a regular `step` grid are accepted, as splitPieces produces them."""
    if not tiles.is_cuda:
        raise RuntimeError("hicdiff_amd.processdata runs on the GPU only: move the tiles to a ROCm device")
    lib = L.load()
    t = tiles.detach().to(torch.float32).contiguous()
    p = t.shape[-1]
    if t.shape[-2] != p or t.numel() != len(origins) * p * p:
        raise ValueError(f"{tuple(tiles.shape)} is not {len(origins)} square tiles")
    org = np.asarray(origins, dtype=np.int64).reshape(-1, 2)
    step = p if step is None else int(step)
    if step < p:
        raise ValueError("step < piece_size: tiles would overlap")
    if len(org) and ((org % step).any() or (org[:, 0] > org[:, 1]).any() or org.min() < 0):
        raise ValueError("origins must be multiples of step with row <= col")
    table = stitch_table(org, n, step)
    nb = table.shape[0]
    out = torch.empty((n, n), dtype=torch.float32, device=t.device)
    if n:
        tb = torch.from_numpy(table).to(t.device)
        src = t if t.numel() else torch.zeros(1, dtype=torch.float32, device=t.device)
        rc = lib.hd_stitch_pieces(_ptr(src), _ptr(tb), nb, p, step, _ptr(out), n, _stream(t))
        if rc != 0:
            raise L.HdError(rc, "hd_stitch_pieces failed")
    return out


def splitPieces(fn, piece_size, step, resol, device="cuda"):
    """Drop-in for the reference's splitPieces: `fn` is the Full_Mats .npy path (an ndarray is accepted too)."""
    data = np.load(fn) if isinstance(fn, (str, os.PathLike)) else np.asarray(fn)
    assert data.shape[0] == data.shape[1]
    if data.dtype != np.float32:
        raise TypeError(f"Full_Mats matrices are float32 (PrepareData_linear_sing.py:176); got {data.dtype}")
    tiles, org = split_pieces_device(torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(device), piece_size, step, resol)
    if len(org) == 0:
        return np.expand_dims(np.asarray([]), 1)                                   # what the reference returns for a too-small map
    return tiles.cpu().numpy().astype(data.dtype, copy=False)


def stitchPieces(tiles, n, piece_size, step, resol, device="cuda"):
    """numpy in / numpy out companion of splitPieces: the (n,n) matrix whose band splitPieces would cut into `tiles`."""
    org, _ = tile_origins(n, piece_size, step, resol)
    t = torch.from_numpy(np.ascontiguousarray(tiles, dtype=np.float32)).to(device)
    return stitch_pieces_device(t, org, n, step).cpu().numpy()


def loadBothConstraints(stria, strib, res):
    raise NotImplementedError("the text-file / cooler leg of the data pipeline is out of scope (SURVEY.md section 8): "
                              "start from DataFull/.../Full_Mats/GSE131811_mat_full_chr_<c>_<res>.npy")


class gse131811Dataset(Dataset):
    """Items (noisy, target, sample, chromosome) from Splits/ (PrepareData_linear_sing.py:225-324)."""

    def __init__(self, full, tvt, res, piece_size, dir, splits, single):
        self.piece_size, self.tvt, self.res, self.full, self.dir = piece_size, tvt, res, full, dir
        table = splits if full else single
        if full and isinstance(tvt, int) and tvt in splits["all"]:
            self.chros = [tvt]
        else:
            self.chros = list(table[tvt])

        def load(kind, c):
            return np.load(f"{dir}/Splits/GSE131811_{kind}_chr_{c}_{res}_piece_{piece_size}.npy")

        first = self.chros[0]
        target, data, samp = load("full", first), load("noisy", first), load("sample", first)
        info = np.repeat(first, target.shape[0])
        for c in self.chros[1:] if full else []:
            t, s, d = load("full", c), load("sample", c), load("noisy", c)
            if len(t):
                target = np.concatenate((target, t))
            if len(s):
                samp = np.concatenate((samp, s))
            if len(d):
                data = np.concatenate((data, d))
                info = np.concatenate((info, np.repeat(c, d.shape[0])))
        self.target, self.data = torch.from_numpy(target), torch.from_numpy(data)
        self.samp, self.info = torch.from_numpy(samp), torch.from_numpy(info)

    def __len__(self):
        return self.target.shape[0]

    def __getitem__(self, idx):
        return self.data[idx], self.target[idx], self.samp[idx], self.info[idx]


class _TileModule:
    """Shared body of the two DataModules; not a LightningDataModule (pytorch_lightning is not a dependency):
    prepare_data / setup / *_dataloader keep their upstream meaning and are called by hand or by train.py."""
    cell_dir = ""
    chromosomes = ()
    ready_count = 0
    splits = {}
    single = {}

    def __init__(self, batch_size=64, res=40000, piece_size=64, cell_line=None, cell_No=1, sigma_0=0.1, deg="deno", channel=1,
                 root=None, device="cuda", num_workers=0):
        self.batch_size, self.res, self.piece_size, self.step = batch_size, res, piece_size, piece_size
        self.cellLine, self.cellNo, self.sigma_0, self.deg, self.channel = cell_line or self.cell_dir, cell_No, sigma_0, deg, channel
        self.root = str(root) if root is not None else os.getcwd()
        self.device, self.num_workers = device, num_workers
        self.dirname = f"{self.root}/DataFull/DataFull_{self.cellLine}_cell{self.cellNo}_{self.res}_{deg}_{sigma_0}"

    def extract_constraint_mats(self):
        raise NotImplementedError("reading .mcool files needs cooler; out of scope (SURVEY.md section 8) -- provide Full_Mats/*.npy")

    def extract_create_numpy(self):
        raise NotImplementedError(f"no {self.dirname}/Full_Mats/GSE131811_mat_full_chr_*_{self.res}.npy found and the "
                                  "cooler/text leg that would create them is out of scope (SURVEY.md section 8)")

    def split_numpy(self):
        os.makedirs(self.dirname + "/Splits", exist_ok=True)
        if not glob.glob(f"{self.dirname}/Full_Mats/GSE131811_mat_full_chr_1_{self.res}.npy"):
            self.extract_create_numpy()
        H_funcs = MakeFunc(deg=self.deg, image_channel=self.channel, image_size=self.piece_size, device="cpu")
        for i in self.chromosomes:
            target = splitPieces(f"{self.dirname}/Full_Mats/GSE131811_mat_full_chr_{i}_{self.res}.npy", self.piece_size, self.step,
                                 resol=self.res, device=self.device)
            stem = f"{self.dirname}/Splits/GSE131811_%s_chr_{i}_{self.res}_piece_{self.piece_size}"
            np.save(stem % "full", target)
            if len(target) == 0:                                                   # a map smaller than nothing: keep the three files consistent
                np.save(stem % "noisy", np.zeros((0, self.channel, self.piece_size, self.piece_size), np.float32))
                np.save(stem % "sample", np.zeros((0, self.channel * self.piece_size ** 2), np.float32))
                continue
            # the degradation of :194-202 stays on the host generator so that a seeded run writes the reference's files
            data_t = torch.from_numpy(target)
            data = H_funcs.H(data_t)
            data = data + self.sigma_0 * torch.randn_like(data)
            pinv_y_0 = H_funcs.H_pinv(data).view(data_t.shape[0], self.channel, self.piece_size, self.piece_size)
            np.save(stem % "noisy", pinv_y_0)
            np.save(stem % "sample", data.numpy())

    def prepare_data(self):
        found = glob.glob(f"{self.dirname}/Splits/GSE131811_full_chr_*_{self.res}_piece_{self.piece_size}.npy")
        if len(found) <= self.ready_count:
            self.split_numpy()

    def _set(self, tvt):
        return gse131811Dataset(True, tvt, self.res, self.piece_size, self.dirname, self.splits, self.single)

    def setup(self, stage=None):
        if isinstance(stage, int) and stage in self.splits["all"]:
            self.test_set = self._set(stage)
        if stage == "fit":
            self.train_set, self.val_set = self._set("train"), self._set("val")
        if stage == "test":
            self.test_set = self._set("test")

    def train_dataloader(self):
        return DataLoader(self.train_set, self.batch_size, num_workers=self.num_workers, shuffle=True)

    def val_dataloader(self):
        return DataLoader(self.val_set, self.batch_size, num_workers=self.num_workers)

    def test_dataloader(self):
        return DataLoader(self.test_set, self.batch_size, num_workers=self.num_workers)


class GSE130711Module(_TileModule):
    """Human single cells (:106-343): chromosomes 1..22, train/val/test split of :237-241."""
    cell_dir = "Human"
    chromosomes = tuple(range(1, 23))
    ready_count = 20
    splits = {"all": tuple(range(1, 23)), "train": (1, 3, 5, 7, 8, 9, 11, 13, 15, 16, 17, 19, 21, 22), "val": (4, 14, 18, 20),
              "test": (2, 6, 10, 12)}
    single = {"train": (15,), "val": (16,), "test": (17,)}


class GSE131811Module(_TileModule):
    """Drosophila single cells (:345-593): chromosomes 1..6, splits of :486-491."""
    cell_dir = "Dros"
    chromosomes = tuple(range(1, 7))
    ready_count = 5
    splits = {"all": tuple(range(1, 7)), "train": (5,), "val": (2,), "test": (1, 2, 3, 4, 5, 6)}
    single = {"train": (5,), "val": (1,), "test": (2,)}
