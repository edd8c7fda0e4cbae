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
    # the only collective of the path, in its single-buffer form: every rank's shard lands at its final offset of ONE
    # [world * per, length, dim] tensor (the list form makes `world` extra device copies of 64 MiB each at dim 512)
    gathered = torch.empty(world * per, length, dim, device=pad.device, dtype=pad.dtype)
    dist.all_gather_into_tensor(gathered, pad)
    if per * world == total:
        return gathered
    parts = []
    for r in range(world):                                       # uneven split: drop the padding rows of the short shards
        rlo, rhi = shard_range(total, r, world)
        parts.append(gathered[r * per: r * per + (rhi - rlo)])
    return torch.cat(parts, dim=0)


# ------------------------------------------------------------------------------------------------ data-parallel training
def collective_report(shard: torch.Tensor, local_ms_per_step: float, reps: int = 3) -> dict:
    """What the sharded sampler's ONE collective costs on this job, measured apart from the sampling loop (the loop
    itself never communicates): the all-gather of every rank's `shard` (generated latents [b, n, d]) in the single-buffer form
    sharded_sample uses, timed `reps` times between barriers (max over ranks, minimum over repetitions), plus the spread of the
    ranks' own loop times.  Every rank must call it; every rank returns the same dict.  With gloo (CPU functional tests) the tensors
    travel through host memory, so the time says nothing about xGMI -- the block says which backend it saw."""
    import time
    world, backend = dist.get_world_size(), dist.get_backend()
    src = shard if backend != "gloo" else shard.cpu()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    times = []
    for _ in range(reps):
        dist.barrier()
        if src.is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(out, src)
        if src.is_cuda:
            torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=src.device)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        times.append(el.item())
    per = torch.zeros(world, dtype=torch.float64, device=src.device)
    per[dist.get_rank()] = local_ms_per_step
    dist.all_reduce(per, op=dist.ReduceOp.SUM)
    nbytes = src.numel() * src.element_size()
    return dict(backend=backend, world=world, collectives_inside_the_loop=0, allgather_ms=round(1e3 * min(times), 4),
                allgather_bytes_per_rank=nbytes, allgather_bytes_received_per_rank=nbytes * (world - 1),
                allgather_gbytes_per_s_per_rank=round(nbytes * (world - 1) / max(min(times), 1e-9) / 1e9, 2),
                per_rank_ms_per_step=dict(min=round(per.min().item(), 4), max=round(per.max().item(), 4),
                                          all=[round(v, 4) for v in per.tolist()]),
                note="backend 'nccl' is RCCL over xGMI; 'gloo' (functional tests on one device / CPU) moves the shards through host memory")


class GradientAllReducer:
    """Data-parallel training step for `NaturalSpeech2.forward` (NS2:1635, NS2:1886: the reference hands this to
    accelerate / DDP): one process per GPU, replicated weights, each rank's loss on its own shard of the batch, gradients averaged
    with bucketed `all_reduce`s (backend "nccl" == RCCL over xGMI) that start WHILE backward is still running.  The backward
    arithmetic underneath is the HIP training path (training.py); this is the collective around it, built for the xGMI topology:

      * xGMI is point-to-point (7 links x ~153 GB/s per GPU) and a ring all-reduce is bound per link, so buckets are LARGE
        (default 64 MiB: the 1.04 GB of fp32 gradients of the d512/L12 model = 17 collectives, each long enough to run at link
        rate), formed in reverse parameter order -- gradients become ready back to front;
      * ONE flat fp32 buffer holds every gradient and every `p.grad` is a VIEW into it: backward accumulates straight into the
        collective's buffer, the averaged result is already where the optimizer reads it -- no per-step concatenation or copy
        back (a `zero_grad(set_to_none=True)` in between is tolerated: the fresh gradient is copied into its view once);
      * a bucket's all-reduce is launched from the post-accumulate hook of its last gradient, but STRICTLY in bucket order: a
        ready bucket waits for the earlier ones, so every rank issues the same collectives in the same order even when ranks
        differ in which parameters received a gradient (whatever is left is issued, in order, by `finish()`);
      * gradient accumulation (`gradient_accumulate_every`, NS2:1877-1885): backward passes inside `with reducer.accumulate():`
        only accumulate; the pass outside it reduces.  A second reducing backward without `finish()` raises instead of silently
        dropping gradients.

        reducer = GradientAllReducer(diffusion.parameters())        # once
        for micro in micro_batches[:-1]:
            with reducer.accumulate():
                diffusion(micro).backward()
        diffusion(micro_batches[-1]).backward()                     # hooks fire the all-reduces during this backward
        reducer.finish()                                            # wait; p.grad now holds the rank-averaged gradient
        optimizer.step(); reducer.zero_grad()
    """

    def __init__(self, parameters, bucket_bytes: int = 64 << 20, process_group=None):
        self.group = process_group
        self.params = [p for p in parameters if p.requires_grad]
        assert self.params, "no parameters to reduce"
        assert all(p.dtype == torch.float32 for p in self.params), "gradients are reduced in fp32"
        self.world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        self._collective = dist.is_initialized()                 # a world of 1 still issues its (degenerate) all-reduces: same code path
        order = list(reversed(self.params))                      # reverse registration order: ready roughly back to front
        self.flat = torch.zeros(sum(p.numel() for p in order), dtype=torch.float32, device=order[0].device)
        self._view, self.buckets, self._range = {}, [], []
        off, cur, cur_bytes, start = 0, [], 0, 0
        for p in order:
            self._view[id(p)] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self.buckets.append(cur)
                self._range.append((start, off))
                cur, cur_bytes, start = [], 0, off
        if cur:
            self.buckets.append(cur)
            self._range.append((start, off))
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        for p in self.params:                                    # adopt gradients that exist already, then point .grad at the views
            if p.grad is not None:
                self._view[id(p)].copy_(p.grad)
            p.grad = self._view[id(p)]
        self._sync = True
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._reset()

    def _reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._launched = 0                                       # buckets [0, _launched) have been issued
        self._armed = True

    # ---- gradient accumulation
    class _Accumulate:
        def __init__(self, owner):
            self.owner = owner

        def __enter__(self):
            self.prev, self.owner._sync = self.owner._sync, False

        def __exit__(self, *exc):
            self.owner._sync = self.prev

    def accumulate(self):
        """context manager: backward passes inside only accumulate into `.grad` (no collective is launched)"""
        return GradientAllReducer._Accumulate(self)

    def zero_grad(self):
        """zero every gradient in place (they stay views of the flat buffer)"""
        self.flat.zero_()
        for p in self.params:
            p.grad = self._view[id(p)]

    def _adopt(self, p):
        v = self._view[id(p)]
        if p.grad is not None and p.grad.data_ptr() != v.data_ptr():     # zero_grad(set_to_none=True) happened: autograd made a new tensor
            v.copy_(p.grad)
            p.grad = v

    def _issue_ready(self):
        while self._launched < len(self.buckets) and self._pending[self._launched] == 0:
            i = self._launched
            if self._collective:
                lo, hi = self._range[i]
                self._work[i] = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._launched += 1

    def _on_grad(self, p):
        self._adopt(p)
        if not self._sync:
            return
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] < 0:
            raise RuntimeError("GradientAllReducer: a second reducing backward ran before finish(); wrap all but the last backward of a "
                               "step in `with reducer.accumulate():` (gradients of the extra pass would be lost)")
        self._issue_ready()

    def finish(self):
        """issue what is left (parameters without a gradient this step contribute their view's content: zero after
        `zero_grad()`), wait for every collective and divide by the world size: `.grad` holds the rank-averaged gradients"""
        for p in self.params:
            if p.grad is None:                                   # never touched since a set_to_none: counts as zero
                self._view[id(p)].zero_()
                p.grad = self._view[id(p)]
            else:
                self._adopt(p)
        for i in range(len(self.buckets)):
            self._pending[i] = 0
        self._issue_ready()
        for w in self._work:
            if w is not None:
                w.wait()
        if self.world > 1:
            self.flat.div_(self.world)
        self._reset()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
