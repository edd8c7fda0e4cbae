#!/usr/bin/env python
"""RVQ encode micro-benchmark at BASELINE config-4 size (32 x 1024 frames, 8 stages x 1024 codes): ms per encode on random and
on all-zero data (same instruction stream: the difference is the power / clock effect), indices checked against the oracle.
NS2_LIB selects an experiment build for same-box A/B."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naturalspeech2_pytorch_amd import ops
from oracle import rvq_oracle as R

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
cb = torch.randn(8, 1024, 128, generator=g)
lat = torch.randn(32 * 1024, 128, generator=g)
out = {}
for name, (c, x) in {"random": (cb, lat), "zeros": (torch.zeros_like(cb), torch.zeros_like(lat))}.items():
    cbd, xd = c.to(dev), x.to(dev)
    norm = ops.rvq_prepare(cbd)
    for _ in range(3):
        codes, emb = ops.rvq_encode(xd, cbd, norm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        codes, emb = ops.rvq_encode(xd, cbd, norm)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    out[name] = dict(ms=round(ms, 4), tflops=round(2.0 * 32768 * 128 * 1024 * 8 / 1e9 / ms, 2))
    if name == "random":
        c_ref, _, _ = R.rvq_encode(lat[:4096], cb)
        out["rows_differing_of_4096"] = int((codes[:4096].cpu() != c_ref).any(dim=-1).sum())
print(json.dumps(dict(lib=os.environ.get("NS2_LIB", "default"), **out)))
