"""Data-parallel sampling over the GPUs of one node: utterances are independent through the whole
sampling loop, so each rank denoises a contiguous shard with NO data-path collective, and a single RCCL
all-gather (backend "nccl" == RCCL over xGMI; "gloo" on CPU for tests) assembles the generated latents.
The reference samples on rank 0 only (NS2:1900-1918); this is new functionality with the same per-utterance result.
"""
import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); returns (rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("NS2_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())   # > 1 rank per device only for functional tests (gloo)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous shard [lo, hi) of `total` utterances for `rank` (first total % world ranks get one extra)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def utterance_noise(lo: int, hi: int, length: int, dim: int, seed: int = 0, device="cpu") -> torch.Tensor:
    """initial noise seeded by the GLOBAL utterance index, so results do not depend on the shard count."""
    out = torch.empty(hi - lo, length, dim)
    for i in range(lo, hi):
        g = torch.Generator().manual_seed(seed * 1000003 + i)
        out[i - lo] = torch.randn(length, dim, generator=g)
    return out.to(device)


def sharded_sample(sample_fn: Callable[[torch.Tensor], torch.Tensor], total: int, length: int, dim: int, seed: int = 0,
                   device="cpu", gather: bool = True) -> torch.Tensor:
    """Each rank runs `sample_fn(noise_shard) -> latents_shard`; one all-gather returns [total, length, dim] on every
    rank (shards padded to equal size for the collective)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(total, rank, world)
    noise = utterance_noise(lo, hi, length, dim, seed, device)
    local = sample_fn(noise) if hi > lo else noise
    if world == 1 or not gather:
        return local
    per = (total + world - 1) // world
    pad = torch.zeros(per, length, dim, device=local.device, dtype=local.dtype)
    pad[: hi - lo] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)                                   # the only collective of the path
    parts = []
    for r in range(world):
        rlo, rhi = shard_range(total, r, world)
        parts.append(bufs[r][: rhi - rlo])
    return torch.cat(parts, dim=0)
