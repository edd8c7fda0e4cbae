import sys, torch
sys.path.insert(0, ".")
from naturalspeech2_pytorch_amd import Model
from oracle import ns2_oracle as O
from tests.golden.gen import make_input, make_weights
DEV = torch.device("cuda:0")
rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm()).item()
for n, b in ((4096, 2), (8192, 1), (1000, 3), (2560, 2)):
    kw = dict(dim=512, depth=2)
    m = Model(**kw, precision="exact")
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5)
    m.load_state_dict(sd); m = m.to(DEV).eval()
    x = make_input("x", (b, n, 512), seed=6); t = make_input("times", (b,), seed=6, uniform=True)
    sdg = {k: v.to(DEV) for k, v in sd.items()}
    with torch.no_grad():
        ref = O.model_forward(sdg, x.to(DEV), t.to(DEV))
        out = {}
        for p in ("exact", "hybrid", "half"):
            m.precision = p
            out[p] = rel(m(x.to(DEV), t.to(DEV)), ref)
    print(n, b, {k: f"{v:.2e}" for k, v in out.items()}, flush=True)
