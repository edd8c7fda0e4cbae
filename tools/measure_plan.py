"""Error of a model precision plan on the MI355X against the fp32 oracle run with its tensors on the GPU (plain PyTorch fp32 ops):
the d512/L12 sweep of tests/test_parity_r2_gpu.py (seeds x diffusion times, per-utterance error, max / mean) and the amplified
feed-forward case (FF-in weights x6, d128/L6).   python tools/measure_plan.py hybrid hybrid_ff [--seeds 4]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import Model  # noqa: E402
from oracle import ns2_oracle as O  # noqa: E402
from tests.golden.gen import make_input, make_weights  # noqa: E402

DEV = torch.device("cuda:0")


def rel_rows(a, b):
    a, b = a.double().flatten(1), b.double().flatten(1)
    return ((a - b).norm(dim=1) / b.norm(dim=1)).tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("plans", nargs="+")
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = {p: {} for p in a.plans}
    times = torch.tensor([0.002, 0.5, 0.999])
    kw = dict(dim=512, depth=12)
    errs = {p: [] for p in a.plans}
    for seed in range(a.seeds):
        shapes = {k: tuple(v.shape) for k, v in Model(**kw).state_dict().items()}
        sd = make_weights(shapes, seed=100 + seed)
        x = make_input("x", (3, 1024, 512), seed=200 + seed)
        with torch.no_grad():
            ref = O.model_forward({k: v.to(DEV) for k, v in sd.items()}, x.to(DEV), times.to(DEV))
        for p in a.plans:
            m = Model(**kw, precision=p)
            m.load_state_dict(sd)
            m = m.to(DEV).eval()
            with torch.no_grad():
                errs[p] += rel_rows(m(x.to(DEV), times.to(DEV)), ref)
            m.check_saturation(sync=True)
            del m
            torch.cuda.empty_cache()
    for p in a.plans:
        res[p]["sweep_d512_L12"] = dict(max=max(errs[p]), mean=sum(errs[p]) / len(errs[p]), n=len(errs[p]))
    # amplified feed-forward branch (tests/test_parity_r2_gpu.py::test_hybrid_plan_with_amplified_ff_branch)
    kw2 = dict(dim=128, depth=6)
    shapes = {k: tuple(v.shape) for k, v in Model(**kw2).state_dict().items()}
    for scale in (6.0, 12.0):
        sd = make_weights(shapes, seed=7)
        sd = {k: (v * scale if k.endswith(".5.0.weight") else v) for k, v in sd.items()}
        x = make_input("x", (4, 1024, 128), seed=8)
        t = torch.tensor([0.1, 0.4, 0.7, 0.95])
        with torch.no_grad():
            ref = O.model_forward({k: v.to(DEV) for k, v in sd.items()}, x.to(DEV), t.to(DEV))
        for p in a.plans:
            m = Model(**kw2, precision=p)
            m.load_state_dict(sd)
            m = m.to(DEV).eval()
            with torch.no_grad():
                e = rel_rows(m(x.to(DEV), t.to(DEV)), ref)
            res[p][f"amplified_ffin_x{int(scale)}_d128_L6"] = max(e)
            del m
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
