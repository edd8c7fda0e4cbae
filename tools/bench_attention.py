#!/usr/bin/env python
"""Self-attention micro-benchmark at the headline shape (B=32, H=8, N=1024, d_head=64): ms per launch and algorithmic TFLOP/s
per precision, plus the relative error against an fp64 softmax(QK^T/8)V.  NS2_LIB selects an experiment build (same-box A/B)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naturalspeech2_pytorch_amd import ops

dev = torch.device("cuda:0")
B, H, N = 32, 8, 1024
g = torch.Generator().manual_seed(3)
q = torch.randn(B * N, H * 64, generator=g)
k = torch.randn(B * N, H * 64, generator=g)
v = torch.randn(B * N, H * 64, generator=g)
out = {}
for prec in (4, 2, 3):
    attp = 2 if prec == 4 else prec                       # attention operands are IEEE half at precision 4
    qp, kp = ops.split(q.to(dev), precision=attp), ops.split(k.to(dev), precision=attp)
    vt = v.reshape(B, N, H * 64).transpose(1, 2).reshape(B * H * 64, N).contiguous()
    vtp = ops.split(vt.to(dev), precision=attp)
    for _ in range(3):
        o = ops.attention(qp, kp, vtp, B, H, N, N, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        o = ops.attention(qp, kp, vtp, B, H, N, N, precision=prec)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    of = ops.join(o, H * 64)[:N].double().cpu()       # utterance 0
    qd, kd, vd = (t[:N].double().reshape(N, H, 64).transpose(0, 1) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(1, 2) / 8.0, dim=-1) @ vd).transpose(0, 1).reshape(N, H * 64)
    out[str(prec)] = dict(ms=round(ms, 4), tflops=round(4.0 * B * H * N * N * 64 / 1e9 / ms, 1),
                          rel_err=float(((of - ref).norm() / ref.norm()).item()))
print(json.dumps(dict(lib=os.environ.get("NS2_LIB", "default"), **out)))
