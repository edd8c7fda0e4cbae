"""Warm training-step timing of the denoiser: forward + loss + backward (+ Adam step) per iteration, HIP training path against the
PyTorch composite on the same GPU.   python tools/bench_train.py [--shapes d128,d512] [--iters 5] [--out file.json]
FLOPs: 3 x the forward's algorithmic FLOPs (26.74 GFLOP / utterance of 1024 frames at d128/L6, 316.37 at d512/L12)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import Model, NaturalSpeech2  # noqa: E402

SHAPES = {"d128": (dict(dim=128, depth=6), 4, 1024, 26.74e9), "d128_b32": (dict(dim=128, depth=6), 32, 1024, 26.74e9),
          "d512": (dict(dim=512, depth=12), 32, 1024, 316.37e9), "d512_b8": (dict(dim=512, depth=12), 8, 1024, 316.37e9)}


def run(name, backend, iters, warm=2, opt_step=True, train_precision="exact", fused_adam=False):
    kw, b, n, gflop_utt = SHAPES[name]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = Model(**kw).to(dev).train()
    m.train_backend = backend
    m.train_precision = train_precision
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000).to(dev)
    # fused=True: PyTorch's single-pass Adam (the default "foreach" form makes ~6 passes over parameters and states)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=fused_adam or None)
    g = torch.Generator(device="cpu").manual_seed(1)
    audio = torch.randn(b, n, kw["dim"], generator=g).to(dev)
    times = torch.rand(b, generator=g).to(dev)
    noise = torch.randn(b, n, kw["dim"], generator=g).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = d(audio, times=times, noise=noise)
        loss.backward()
        if opt_step:
            opt.step()
        return loss

    for _ in range(warm):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    flops = 3 * gflop_utt * b * n / 1024
    return dict(shape=name, backend=backend, train_precision=train_precision, fused_adam=fused_adam, batch=b, frames=n, ms_per_step=ms,
                loss=float(loss), algorithmic_tflops=flops / (ms * 1e-3) / 1e12, peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="d128,d512")
    ap.add_argument("--backends", default="hip,composite")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--train-precision", default="exact", help="comma separated: exact (bf16 x3), mixed (half product + fp8 terms, loss-scaled)")
    ap.add_argument("--fused-adam", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = []
    for s in a.shapes.split(","):
        for bk in a.backends.split(","):
            for tp in (a.train_precision.split(",") if bk == "hip" else ["exact"]):
                torch.cuda.reset_peak_memory_stats()
                try:
                    r = run(s, bk, a.iters, train_precision=tp, fused_adam=a.fused_adam)
                except Exception as e:                              # noqa: BLE001
                    r = dict(shape=s, backend=bk, train_precision=tp, error=repr(e)[:300])
                print(json.dumps(r), flush=True)
                res.append(r)
                torch.cuda.empty_cache()
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
