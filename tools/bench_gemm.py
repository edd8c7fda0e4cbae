"""Micro-benchmark of the GEMM family at the headline shapes (for rocprofv3 --pmc passes and A/B of kernel variants).
    python tools/bench_gemm.py [--prec 3|4|2|1] [--which ffconv|ffin|ffout|qkv|wavenet|all] [--iters 20] [--kernel 0|1|2]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prec", type=int, default=3)
ap.add_argument("--which", default="all")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--kernel", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")
_lib.check(_lib.load().ns2_debug_force_gemm(args.kernel))
B, N, d, f = 32, 1024, 512, 1365
M = B * N
g = torch.Generator().manual_seed(0)


def rnd(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(dev)


cases = {}
fp = ops.round_up(f, 32)
x512 = ops.split(rnd(M, d), precision=args.prec)
xf = ops.split(rnd(M, f), ldo=fp, precision=args.prec)
if args.which in ("all", "ffconv"):
    w = ops.PackedWeight(rnd(f, f, 3, scale=0.02), precision=args.prec); b = rnd(f)
    cases["ffconv  [M,1365]x[1365,3x1365]"] = (lambda: ops.linear_split(w, xf, bias=b, conv_taps=3, dilation=1, seq_len=N, precision=args.prec), 2.0 * M * f * 3 * f)
if args.which in ("all", "ffin"):
    w1 = ops.PackedWeight(rnd(2 * f, d, scale=0.04), geglu=True, precision=args.prec); pb = ops.geglu_pack_bias(rnd(2 * f), f)
    cases["ffin+geglu [M,512]x[512,2730]"] = (lambda: ops.linear_geglu(w1, x512, pb, precision=args.prec), 2.0 * M * d * 2 * f)
if args.which in ("all", "ffout"):
    w2 = ops.PackedWeight(rnd(d, f, scale=0.03), precision=args.prec); b2 = rnd(d); r = rnd(M, d)
    cases["ffout+res [M,1365]x[1365,512]"] = (lambda: ops.linear_f32(w2, xf, bias=b2, resid=r, precision=args.prec), 2.0 * M * f * d)
if args.which in ("all", "qkv"):
    wq = ops.PackedWeight(rnd(1536, d, scale=0.04), precision=args.prec)
    cases["qkv [M,512]x[512,1536]"] = (lambda: ops.linear_qkv(wq, x512, seq_len=N, split_col=1024, precision=args.prec), 2.0 * M * d * 1536)
if args.which in ("all", "outproj"):
    wo = ops.PackedWeight(rnd(d, d, scale=0.04), precision=args.prec); ro = rnd(M, d)
    cases["outproj+res [M,512]x[512,512]"] = (lambda: ops.linear_f32(wo, x512, resid=ro, precision=args.prec), 2.0 * M * d * d)
if args.which in ("all", "wavenet"):
    ww = ops.PackedWeight(rnd(d, d, 3, scale=0.03), extra1x1=rnd(d, d, 1, scale=0.04), precision=args.prec); bc, br = rnd(d), rnd(d); film = rnd(B, 2 * d)
    cases["wavenet block dil=16 [M,512]x[512,4x512]"] = (lambda: ops.wavenet_block(ww, x512, N, 16, bc, br, film, precision=args.prec), 2.0 * M * d * 4 * d)

for name, (fn, flops) in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f"{name:45s} prec={args.prec} kernel={args.kernel}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s algorithmic  ({flops*{3: 3, 4: 2}.get(args.prec, 1)/ms/1e9:7.1f} in 16-bit MFMA units)")
