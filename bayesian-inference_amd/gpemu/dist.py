"""One process per GPU: process-group bootstrap and the few host-side exchanges the drop-in modules need.

The reference's only parallel mechanism is a multiprocessing pool inside ``run_mcmc`` (ref: mcmc.py:77-85);
``steer_analysis.py`` never initialises a process group.  Launched as
``python -m torch.distributed.run --nproc-per-node N -m bayesian_inference.steer_analysis ...`` the drop-in
modules therefore join the group themselves, from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK /
MASTER_ADDR / MASTER_PORT), the first time they ask for their rank -- before any GPU work of the stage.
"""
from __future__ import annotations

import os

import numpy as np


def _env_world():
    """WORLD_SIZE of a launcher's environment; 1 when absent, empty or not a number (some schedulers export it so:
    a malformed value must not break a plain single-process run)."""
    try:
        return max(1, int(os.environ.get("WORLD_SIZE", "1")))
    except ValueError:
        return 1


def _dist():
    """torch.distributed, or None where there cannot be a process group: torch absent -- or not imported by anybody yet in
    a process that no launcher started (importing it costs ~1 s, a third of a C3 `fit_emulators`).  So several ranks
    are seen only if torch is imported, or RANK and WORLD_SIZE are set, BEFORE the first sampler / fit call asks for its
    rank ("Multi-GPU")."""
    import sys
    if "torch" not in sys.modules and (_env_world() <= 1 or "RANK" not in os.environ):
        return None
    try:
        import torch.distributed as dist
        return dist if dist.is_available() else None
    except ImportError:
        return None


def ensure_process_group():
    """Join the launcher's process group if there is one and nobody has yet.  Backend: RCCL ("nccl") with
    the device taken from LOCAL_RANK when a GPU is visible, else gloo; GPEMU_DIST_BACKEND overrides
    (tests run several ranks on one GPU over gloo)."""
    dist = _dist()
    if dist is None or dist.is_initialized():
        return
    world = _env_world()
    if world <= 1 or "RANK" not in os.environ:
        return
    import torch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    rank = int(os.environ["RANK"])
    backend = os.environ.get("GPEMU_DIST_BACKEND")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)


def rank_world():
    """(rank, world size) of this process; (0, 1) outside torch.distributed."""
    ensure_process_group()
    dist = _dist()
    if dist is not None and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def rank0_array(arr, group=None):
    """Rank 0's copy of ``arr`` (same shape and dtype on every rank) on every rank."""
    dist = _dist()
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return arr
    import torch
    where = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    buf = torch.from_numpy(np.ascontiguousarray(arr).copy()).to(where)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(buf, src=src, group=group)
    return buf.cpu().numpy()


def rank0_int(value, group=None):
    """Rank 0's integer on every rank."""
    return int(rank0_array(np.array([int(value)], dtype=np.int64), group)[0])
