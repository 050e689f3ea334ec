"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The particle batch is the only sharded axis.  Per training step there are exactly two collectives:
  1. forward : all-reduce(SUM) of ONE flat fp32 buffer [raw histogram sums of every projection | sum log_prob |
               sum |x|^2]  — before the (non-linear) normalisation / discrepancy, which every rank then evaluates
               redundantly on the tiny reduced tensor, so L, H, D are identical on all ranks;
  2. backward: all-reduce(SUM) of the flat parameter-gradient vector (158 890 floats for the 6-D NSF).
Both are latency-bound (25.6 KB .. 2.9 MB and 636 KB); no per-particle data ever crosses xGMI.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def is_active() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size() -> int:
    return dist.get_world_size() if is_active() else 1


def rank() -> int:
    return dist.get_rank() if is_active() else 0


def init_from_env(backend: Optional[str] = None, seed: Optional[int] = None) -> torch.device:
    """Initialise from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); returns this rank's device."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    # rehearsal on a one-GPU box: MENTFLOW_SHARE_GPU=1 puts every rank on cuda:0 and talks gloo (RCCL refuses two ranks
    # on one device); the production path is one rank per GPU over RCCL
    share = use_gpu and os.environ.get("MENTFLOW_SHARE_GPU") == "1"
    if share:
        local, backend = 0, backend or "gloo"
    if use_gpu:
        torch.cuda.set_device(local)
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if use_gpu else "gloo")
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    if seed is not None:
        torch.manual_seed(seed + rank())      # every rank draws its own particles
    return device


def local_batch(global_batch: int) -> int:
    """Particles this rank samples out of a global batch (remainder goes to the low ranks)."""
    w, r = world_size(), rank()
    return global_batch // w + (1 if r < global_batch % w else 0)


class AllReduceSumFn(torch.autograd.Function):
    """y = sum over ranks of x.  Every rank goes on to compute the SAME scalar loss from y, and the per-rank backward
    passes are later summed by the gradient all-reduce, so the adjoint w.r.t. the local contribution is the
    identity (not another all-reduce)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor) -> torch.Tensor:
        y = x.detach().clone()
        if is_active():
            dist.all_reduce(y, op=dist.ReduceOp.SUM)
        return y

    @staticmethod
    def backward(ctx, gy):
        return gy


def all_reduce_sum(x: torch.Tensor) -> torch.Tensor:
    return AllReduceSumFn.apply(x) if is_active() else x


def reduce_gradients_(flat_grad: torch.Tensor) -> None:
    if is_active():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)


def barrier() -> None:
    if is_active():
        dist.barrier()
