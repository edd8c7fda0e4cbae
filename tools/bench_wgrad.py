"""Weight-gradient GEMM of the training step, per shape of the d512 / L12 model at 32 x 1024 tokens: the transposed-copy route
(ns2_grad_prep / ns2_planes_transpose + ns2_wgrad) against the row-plane route (ns2_wgrad_rows: LDS transpose reads), each part timed.
    python tools/bench_wgrad.py [--precision 4] [--iters 10]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import training  # noqa: E402

SHAPES = [("out_proj / wavenet 1x1", 512, 512, 1, 1), ("wavenet dilated conv", 512, 512, 3, 4), ("qkv", 1536, 512, 1, 1), ("ff_in", 2730, 512, 1, 1),
          ("ff_out", 512, 1365, 1, 1), ("ff_conv", 1365, 1365, 3, 1)]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tokens", type=int, default=32768)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    hb = training.HipBackend(a.precision)
    M, seq = a.tokens, 1024
    g = torch.Generator().manual_seed(0)
    for name, R, K, T, dil in SHAPES:
        if a.only and a.only not in name:
            continue
        Kp = (K + 31) // 32 * 32
        dy = (torch.randn(M, R, generator=g) * 0.5).to(dev)
        x = torch.randn(M, Kp, generator=g)
        x[:, K:] = 0
        xp = hb.split(x.to(dev))
        shifts = tuple((T - 1 - t) * dil for t in range(T))
        row, dyt, _ = hb.grad_prep(dy, R, want_row=True, want_t=True)
        xt = hb.transpose(xp, 0, Kp, seq if T > 1 else 0, shifts)
        res = dict(shape=name, R=R, K=K, T=T, tokens=M, precision=a.precision, gflop=2.0 * R * K * T * M / 1e9)
        res["prep_row_and_t_us"] = timed(lambda: hb.grad_prep(dy, R, want_row=True, want_t=True), a.iters)
        res["prep_row_only_us"] = timed(lambda: hb.grad_prep(dy, R, want_row=True), a.iters)
        res["transpose_x_us"] = timed(lambda: hb.transpose(xp, 0, Kp, seq if T > 1 else 0, shifts), a.iters)
        res["wgrad_transposed_us"] = timed(lambda: hb.wgrad(dyt, xt, R, T, K), a.iters)
        res["wgrad_rows_us"] = timed(lambda: hb.wgrad_rows(row, xp, R, T, K, dil, seq), a.iters)
        res["route_transposed_us"] = res["prep_row_and_t_us"] + res["transpose_x_us"] + res["wgrad_transposed_us"]
        res["route_rows_us"] = res["prep_row_only_us"] + res["wgrad_rows_us"]
        res["tflops_transposed"] = res["gflop"] / res["wgrad_transposed_us"] / 1e3
        res["tflops_rows"] = res["gflop"] / res["wgrad_rows_us"] / 1e3
        print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)
        del dy, x, xp, row, dyt, xt
        torch.cuda.empty_cache()
