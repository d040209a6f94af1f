"""Tile sharding across the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no multi-GPU path.  Tiles are independent for the whole reverse
chain -- GroupNorm is per sample, attention is within a tile, noise is keyed by the global tile index
-- so the batch is cut into contiguous rank slices, every rank runs its own T-step loop with no
communication, and ONE all-gather of the finished tiles (B/N * S*S * 4 bytes per rank; 512 KiB at
B=256, S=64, N=8) reassembles them in rank order.  Uneven slices are padded to the largest slice for
the collective and trimmed afterwards.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch


def shard_range(n_tiles: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) slice of rank `rank`; the first n % world ranks get one extra tile."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, extra = divmod(n_tiles, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_tiles(local: torch.Tensor, dist, n_tiles: int = None) -> torch.Tensor:
    """Gather equally shaped (b,1,S,S) slices from every rank into (world*b,1,S,S), rank-ordered.
    With n_tiles given, slices may be ragged (shard_range layout): they are padded to the largest
    slice for the collective and the padding rows are dropped."""
    world = dist.get_world_size()
    if n_tiles is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    sizes = [shard_range(n_tiles, r, world) for r in range(world)]
    bmax = max(e - s for s, e in sizes)
    padded = torch.zeros((bmax,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    padded[: local.shape[0]] = local
    out = torch.empty((world * bmax,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(out, padded)
    parts: List[torch.Tensor] = [out[r * bmax: r * bmax + (e - s)] for r, (s, e) in enumerate(sizes)]
    return torch.cat(parts, dim=0)


def sample_sharded(run_local: Callable[[int, int], torch.Tensor], n_tiles: int, dist=None) -> torch.Tensor:
    """Run `run_local(start, count)` on this rank's slice and all-gather the results.

    `run_local` must key its noise by the global tile index (e.g. set ``diffusion.tile_offset = start``
    before ``diffusion.sample``), which makes the gathered result independent of the rank count."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return run_local(0, n_tiles)
    start, stop = shard_range(n_tiles, dist.get_rank(), dist.get_world_size())
    local = run_local(start, stop - start)
    return all_gather_tiles(local, dist, n_tiles)


def sample_tiles(diffusion, n_tiles: int, cond: torch.Tensor = None, dist=None) -> torch.Tensor:
    """Sharded ``diffusion.sample`` / ``super_resolution`` over n_tiles tiles (cond: all tiles, every rank)."""
    def run_local(start, count):
        diffusion.tile_offset = start
        dev = diffusion.betas.device
        if cond is not None:
            return diffusion.super_resolution(cond[start:start + count].to(dev))
        return diffusion.sample(torch.zeros(count, 1, diffusion.image_size, diffusion.image_size))
    try:
        return sample_sharded(run_local, n_tiles, dist)
    finally:
        diffusion.tile_offset = 0


def launch_ranks(argv: List[str], n: int, timeout: float = None, extra_env: dict = None) -> int:
    """Start `n` rank processes of ``argv`` on this node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in the environment, as torch.distributed.run sets them) and wait for them.

    The caller must not have touched the GPU: children are fresh processes (`subprocess.Popen`), nothing is
    exec'd over an initialised one.  Children inherit stdout / stderr, so the one line rank 0 prints is the
    caller's output.  Returns 0 when every rank exits 0; if any rank fails (or `timeout` seconds pass) the
    remaining ranks -- which would otherwise wait in a collective forever -- are terminated and the first
    non-zero code (124 for a timeout) is returned."""
    import os
    import socket
    import subprocess
    import time as _time
    if n < 1:
        raise ValueError("need at least one rank")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env))
    t0, rc = _time.monotonic(), 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if timeout is not None and _time.monotonic() - t0 > timeout:
                rc = 124
                break
            _time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc
