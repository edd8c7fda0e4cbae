"""Deterministic weight / input generators shared by tests, bench parity legs and make_golden.py.

Weights are NOT the reference's RNG stream (that cannot be reproduced without the reference); they are
drawn from a CPU torch.Generator per key so that the same tensors can be rebuilt on the GPU box and
loaded into both the reference `Model` (in make_golden.py) and the HIP `Model`.
"""
import math
import os
import zlib

import torch


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
    return g


def _numel(shp):
    n = 1
    for d in shp:
        n *= d
    return n


def make_weights(shapes: dict, seed: int = 0) -> dict:
    """shapes: {state_dict key: shape}. Linear/conv weights ~ N(0, 1/fan_in); biases ~ 0.1 N(0,1);
    gammas ~ 1 + 0.1 N; sinusoid freqs ~ N(0,1); latents / null tokens ~ 0.3 N."""
    # every key draws from its own generator, so the keys can be drawn in parallel (torch.randn releases the GIL): the 100 M weights of the
    # headline model took 5-6 s of one core per test that builds it
    keys = sorted(shapes)
    big = [k for k in keys if _numel(shapes[k]) >= (1 << 16)]
    drawn = {}
    if len(big) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            for k, r in zip(big, ex.map(lambda kk: torch.randn(tuple(shapes[kk]), generator=_gen(seed, kk)), big)):
                drawn[k] = r
    sd = {}
    for k in keys:
        shp = tuple(shapes[k])
        r = drawn.pop(k) if k in drawn else torch.randn(shp, generator=_gen(seed, k))
        if k.endswith("gamma"):
            w = 1.0 + 0.1 * r
        elif k.endswith("bias"):
            w = 0.1 * r
        elif k.endswith("to_time_cond.0.weights"):
            w = r
        elif k.endswith("weight") and len(shp) >= 2:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            w = r / math.sqrt(fan_in)
        else:
            w = 0.3 * r
        sd[k] = w.contiguous()
    return sd


def make_input(name: str, shape, seed: int = 0, uniform: bool = False) -> torch.Tensor:
    g = _gen(seed, "input:" + name)
    if uniform:
        return torch.rand(tuple(shape), generator=g)
    return torch.randn(tuple(shape), generator=g)
