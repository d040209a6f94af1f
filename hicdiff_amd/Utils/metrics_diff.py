"""Evaluation harness of the unsupervised (DDRM) path: drop-in for ``src/Utils/metrics_diff.py``.

``VisionMetrics(...).getMetrics(model, ...)`` walks a test split of the Splits/ files through the DataModule
(``processdata``), denoises every batch with ``efficient_generalized_steps`` on the HIP engine, and writes
``Outputs_diff/<model><cell><N>_<deg>_<sigma>_trans2_<timestep>/{predict,target,noisy,inds}.npy`` -- ``inds`` holding the
CHROMOSOME of every tile (src/Utils/metrics_diff.py:121-224; processdata/PrepareData_linear_sing.py:323-324).
The tile-quality numbers of the loop (src/Utils/stard_metrics.py:146-160) come from ``hd_tile_metrics`` on the GPU.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from ..functions.denoising import efficient_generalized_steps
from ..functions.H_func import MakeFunc
from .loss.SSIM import ssim
from .metrics import MetricLog


def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    """src/Utils/metrics_diff.py:36-81: numpy fp64 tables ('sigmoid': the torch fp64 form of hicdiff.py, returned as fp32)."""
    n = num_diffusion_timesteps
    if beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, n, dtype=np.float64)
    elif beta_schedule == "const":
        betas = beta_end * np.ones(n, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(n, 1, n, dtype=np.float64)
    elif beta_schedule == "sigmoid":
        t = torch.linspace(0, n, n + 1, dtype=torch.float64) / n
        v_start, v_end = torch.tensor(-3.0).sigmoid(), torch.tensor(3.0).sigmoid()
        ac = (-((t * 6 - 3)).sigmoid() + v_end) / (v_end - v_start)
        ac = ac / ac[0]
        betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999).float()
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (n,)
    return betas


def _data_module(cell_line, cellN, deg, sigma, image_size, root, batch_size=64):
    """The reference's choice of DataModule (:128-138): population data for cell 1 / 22, single cells for 2..6."""
    if cellN == 1 or cellN == 22:
        from ..processdata.PrepareData_linear import GSE130711Module as human, GSE131811Module as dros
    elif cellN in (2, 3, 4, 5, 6):
        from ..processdata.PrepareData_linear_sing import GSE130711Module as human, GSE131811Module as dros
    else:
        raise ValueError(f"cellN = {cellN}: the reference knows cells 1, 22 (population) and 2..6 (single)")
    cls = {"Dros": dros, "Human": human}.get(cell_line)
    if cls is None:
        raise ValueError(f"cell_line = {cell_line!r}: expected 'Human' or 'Dros'")
    return cls(batch_size=batch_size, piece_size=image_size, deg=deg, sigma_0=sigma, cell_No=cellN, root=root)


class VisionMetrics:
    def __init__(self, image_channel=1, image_size=64, sehedule='linear', timestep=20):
        self.ssim = ssim
        self.metric_logs = {"pas_pcc": [], "pas_spc": [], "pas_psnr": [], "pas_ssim": [], "pas_mse": [], "pas_snr": [], "pas_gds": []}
        betas = get_beta_schedule(beta_schedule=sehedule, beta_start=0.0001, beta_end=0.02, num_diffusion_timesteps=1000)
        self.betas = torch.from_numpy(betas).float() if sehedule == 'linear' else betas
        self.num_timesteps = betas.shape[0]
        self.image_channel, self.image_size, self.timestep = image_channel, image_size, timestep
        self.seed = 1234                 # device Philox key of x_T and the per-step draws (the reference uses torch's global generator)
        self.noise = None                # tests: an object with .randn(shape) replaying the reference's draws
        self.last_result = None          # the running test_result of the loop just finished

    def log_means(self, name):
        return (name, np.mean(self.metric_logs[name]))

    def getMetrics(self, model, model_name='HiCdiff', device=None, chro="test", deg='deno', sigma=0.1, cellN=21, cell_line="Dros_cell",
                   res=None, root=None, outdir=None):
        """``model``: the epsilon-network module (``diffusion.model``, inference.py:109).  ``root``: directory holding DataFull/
        (default: the working directory, where the reference's pyrootutils root would be); ``outdir``: where Outputs_diff/ goes."""
        device = torch.device(device) if device is not None else next(model.parameters()).device
        root = os.getcwd() if root is None else str(root)
        dm_test = _data_module(cell_line, cellN, deg, sigma, self.image_size, root)
        dm_test.prepare_data()
        dm_test.setup(stage=chro)
        test_loader = dm_test.test_dataloader()
        H_funcs = MakeFunc(deg=deg, image_channel=self.image_channel, image_size=self.image_size, device=device)

        Outdir = os.path.join(outdir if outdir is not None else root, "Outputs_diff")
        ModelResult = model_name + cell_line + str(cellN) + "_" + deg + "_" + str(sigma) + "_trans2_" + str(self.timestep)
        os.makedirs(os.path.join(Outdir, ModelResult), exist_ok=True)

        log = MetricLog()
        pr, hrs, lrs, indss = [], [], [], []
        seen = 0
        with torch.no_grad():
            for lr, hr, sp, inds in test_loader:
                sp, hr = sp.to(device), hr.to(device)
                n = sp.shape[0]
                if self.noise is not None:
                    x = self.noise.randn((n, self.image_channel, self.image_size, self.image_size)).to(device)
                else:
                    x = model.engine(device).randn(n, self.image_size, self.seed, seen, 1 << 20)
                out, _ = self.sample_image(x, model, H_funcs, sp, sigma, device=device, last=False, tile_offset=seen)
                out = out[-1]
                pr.append(out.cpu()); hrs.append(hr.cpu()); lrs.append(lr); indss.append(inds)
                log.update(out, hr)                                       # on [0,1]-rescaled tiles, as stard_metrics does
                seen += n
        predict = torch.cat(pr).numpy() if pr else np.zeros((0, self.image_channel, self.image_size, self.image_size), np.float32)
        base = os.path.join(Outdir, ModelResult)
        np.save(os.path.join(base, "target"), torch.cat(hrs).numpy() if hrs else predict)
        np.save(os.path.join(base, "noisy"), torch.cat(lrs).numpy() if lrs else predict)
        np.save(os.path.join(base, "predict"), predict)
        np.save(os.path.join(base, "inds"), torch.cat(indss).numpy() if indss else np.zeros((0,), np.int64))
        self.last_result, self.last_dir = dict(log.r), base
        if log.r["nsamples"]:
            for k, v in (("pas_pcc", "pcc"), ("pas_psnr", "psnr"), ("pas_ssim", "ssim"), ("pas_snr", "snr")):
                self.metric_logs[k].append(log.r[v])
            self.metric_logs["pas_mse"].append(log.r["mse"] / log.r["nsamples"])
        return predict

    def sample_image(self, x, model, H_funcs, y_0, sigma_0, device=None, last=False, cls_fn=None, classes=None, tile_offset=0):
        """:215-224: `timestep` of the 1000 steps, eta_B = 1, eta_A = eta_C = 0.85."""
        skip = self.num_timesteps // self.timestep
        seq = range(0, self.num_timesteps, skip)
        self.betas = self.betas.to(device if device is not None else x.device)
        x = efficient_generalized_steps(x, seq, model, self.betas, H_funcs, y_0, sigma_0, etaB=1.0, etaA=0.85, etaC=0.85, cls_fn=cls_fn,
                                        classes=classes, device=device, noise=self.noise, seed=self.seed, tile_offset=tile_offset)
        if last:
            x = x[0][-1]
        return x
