"""Evaluation harness of the supervised (conditional / SR3) path: drop-in for ``src/Utils/metrics_cond.py``.

``VisionMetrics(...).getMetrics(model, ...)``: ``model`` is the callable the reference passes (``diffusion.super_resolution``,
inference.py:131-137); every batch of the test split's low-coverage tiles goes through it and the results are written to
``Outputs_diff/<model><cell><N>_<deg>_<sigma>_test_<type>/{predict,target,noisy,inds}.npy`` (:61-137).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .loss.SSIM import ssim
from .metrics import MetricLog
from .metrics_diff import _data_module


class VisionMetrics:
    def __init__(self, image_channel=1, image_size=64, timestep=1000, type='condition'):
        self.ssim = ssim
        self.metric_logs = {"pas_pcc": [], "pas_spc": [], "pas_psnr": [], "pas_ssim": [], "pas_mse": [], "pas_snr": [], "pas_gds": []}
        self.image_channel, self.image_size, self.timestep, self.type = image_channel, image_size, timestep, type
        self.last_result = None

    def log_means(self, name):
        return (name, np.mean(self.metric_logs[name]))

    def getMetrics(self, model, model_name='HiCdiff', device=None, chro="test", deg='deno', sigma=0.1, cellN=21, cell_line="Dros_cell",
                   root=None, outdir=None):
        owner = getattr(model, "__self__", None)                          # the diffusion object when a bound method was passed
        if device is None:
            device = owner.betas.device if owner is not None and hasattr(owner, "betas") else torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        root = os.getcwd() if root is None else str(root)
        dm_test = _data_module(cell_line, cellN, deg, sigma, self.image_size, root)
        dm_test.prepare_data()
        dm_test.setup(stage=chro)
        test_loader = dm_test.test_dataloader()

        Outdir = os.path.join(outdir if outdir is not None else root, "Outputs_diff")
        ModelResult = model_name + cell_line + str(cellN) + "_" + deg + "_" + str(sigma) + "_test_" + self.type
        os.makedirs(os.path.join(Outdir, ModelResult), exist_ok=True)

        log = MetricLog()
        pr, hrs, lrs, indss = [], [], [], []
        seen = 0
        with torch.no_grad():
            for lr, hr, _, inds in test_loader:
                lr, hr = lr.to(device), hr.to(device)
                if owner is not None and hasattr(owner, "tile_offset"):
                    owner.tile_offset = seen                              # device noise keyed by the tile's position in the test set
                out = model(lr)
                pr.append(out.cpu()); hrs.append(hr.cpu()); lrs.append(lr.cpu()); indss.append(inds)
                log.update(out, hr)
                seen += lr.shape[0]
        if owner is not None and hasattr(owner, "tile_offset"):
            owner.tile_offset = 0
        predict = torch.cat(pr).numpy() if pr else np.zeros((0, self.image_channel, self.image_size, self.image_size), np.float32)
        base = os.path.join(Outdir, ModelResult)
        np.save(os.path.join(base, "target"), torch.cat(hrs).numpy() if hrs else predict)
        np.save(os.path.join(base, "noisy"), torch.cat(lrs).numpy() if lrs else predict)
        np.save(os.path.join(base, "predict"), predict)
        np.save(os.path.join(base, "inds"), torch.cat(indss).numpy() if indss else np.zeros((0,), np.int64))
        self.last_result, self.last_dir = dict(log.r), base
        return predict
