"""Round 6 experiment 1: does the 128-B alignment of operand rows matter to the FF causal conv?
The hybrid plan's conv reads DENSE IEEE-half rows of 1376 elements = 2752 B = 21.5 cache lines: every odd row's 128-B K tile
straddles two lines (12 instead of 8 line requests per 1 KiB LDS-DMA instruction), and so does every odd (row + tap) of W.
Same product with the channel dimension zero-padded to 1408 (22 lines per row, all tiles aligned, +2.3 % MFMA work):
    python tools/exp_align.py [--iters 30]
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, N, f = 32, 1024, 1365
M = B * N
g = torch.Generator().manual_seed(0)
x = torch.randn(M, f, generator=g)
w = torch.randn(f, f, 3, generator=g) * 0.02
b = torch.randn(f, generator=g).to(dev)
flops = 2.0 * M * f * 3 * f
cases = {}
for prec in (2, 4):
    for Kp in (1376, 1408, 1472):
        xp = torch.zeros(M, Kp); xp[:, :f] = x
        wp = torch.zeros(f, Kp, 3); wp[:, :f, :] = w
        a = ops.split(xp.to(dev), ldo=Kp, precision=prec)
        pw = ops.PackedWeight(wp.to(dev), precision=prec)
        cases[(prec, Kp)] = (a, pw)
outs = {}
def run(k):
    a, pw = cases[k]
    return ops.linear_split(pw, a, bias=b, conv_taps=3, dilation=1, seq_len=N, precision=k[0])
for k in cases:
    outs[k] = ops.join(run(k), f).cpu()
for prec in (2, 4):
    ref = outs[(prec, 1376)]
    for Kp in (1408, 1472):
        print(f"prec {prec} Kp {Kp}: bit-identical to Kp 1376: {torch.equal(ref, outs[(prec, Kp)])}, max diff {(ref - outs[(prec, Kp)]).abs().max().item():.3e}")
res = {k: [] for k in cases}
for r in range(args.rounds):
    for k in cases:
        for _ in range(3):
            run(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run(k)
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / args.iters)
for k, v in res.items():
    v = sorted(v); med = v[len(v) // 2]
    print(f"prec {k[0]} Kp {k[1]}: median {med*1e3:7.1f} us  min {v[0]*1e3:7.1f} us   {flops/med/1e9:7.1f} TF algorithmic (median)")
