#!/usr/bin/env python
"""Where does the conditioned model's rounding error enter?  Per-stage taps of the HIP Model (conditioned d512/L12, torch
default init like bench.py) against the fp32 oracle, per precision mode.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naturalspeech2_pytorch_amd import Model
from oracle import ns2_oracle as O

dev = torch.device("cuda:0")
dim, depth = int(os.environ.get("DIM", 512)), int(os.environ.get("DEPTH", 12))
torch.manual_seed(1234)
m = Model(dim=dim, depth=depth, dim_prompt=512, condition_on_prompt=True).to(dev).eval()
sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(5)
b, n = 1, 256
x, t = torch.randn(b, n, dim, generator=g), torch.rand(b, generator=g)
p, c = torch.randn(b, 40, 512, generator=g), torch.randn(b, 512, n, generator=g)
otaps = {}
with torch.no_grad():
    ref = O.model_forward(sd, x, t, p, c, taps=otaps)
names = {"t": b * m.dim * 4 if False else None}
Tc = otaps["t"].numel()
want = {"t": Tc, "c": otaps["c"].numel(), "wavenet.init": b * n * dim, "wavenet.out": b * n * dim}
for i in range(depth):
    want[f"layer{i}.attn"] = b * n * dim
    want[f"layer{i}"] = b * n * dim
def rel(a, r):
    a, r = a.double().cpu().reshape(-1), r.double().reshape(-1)
    return ((a - r).norm() / r.norm()).item()
for prec in ("exact", "mixed", "hybrid", "half"):
    m.precision = prec
    with torch.no_grad():
        out, got = m.debug_forward(x.to(dev), t.to(dev), dict(want), prompt=p.to(dev), cond=c.to(dev))
    row = [f"out {rel(out, ref):.2e}", f"t {rel(got['t'], otaps['t']):.1e}", f"c {rel(got['c'], otaps['c']):.2e}",
           f"wn.init {rel(got['wavenet.init'], otaps['wavenet.init']):.2e}", f"wn.out {rel(got['wavenet.out'], otaps['wavenet.out']):.2e}"]
    row += [f"L{i} {rel(got[f'layer{i}'], otaps[f'transformer.layer{i}']):.2e}" for i in range(depth)]
    print(prec, " ".join(row), flush=True)
