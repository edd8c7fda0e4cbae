"""BASELINE config 1's sampling half at its stated size: Model(dim=128, depth=6), timesteps=1000, sample(length=1024), batch 1 --
wall time of the 1000-step run with the plain loop and with HIP-graph replay of the step."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import Model, NaturalSpeech2

dev = torch.device("cuda:0")
res = {}
for prec in ("hybrid", "exact"):
    torch.manual_seed(0)
    m = Model(dim=128, depth=6, precision=prec).to(dev).eval()
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=1000).to(dev)
    noise = torch.randn(1, 1024, 128)
    for graph in (False, True):
        d.sample(length=1024, noise=noise, use_graph=graph)          # warm-up (packs, graph capture path)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = d.sample(length=1024, noise=noise, use_graph=graph)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[f"{prec}/{'graph' if graph else 'loop'}"] = dict(s=min(ts), steps_per_s=1000 / min(ts), all=ts)
    a = d.sample(length=1024, noise=noise, use_graph=False)
    b = d.sample(length=1024, noise=noise, use_graph=True)
    res[f"{prec}/graph_equals_loop"] = bool(torch.equal(a, b))
print(json.dumps(res, indent=1))
