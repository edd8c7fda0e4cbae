"""Small batches: Model(dim, depth) forward steps at B = 1 ... 32 utterances of 1024 frames with the automatic dispatch (split-K of
the small products, workspace scratch lent by ns2_model_forward), with the split switched off (ns2_debug_force_gemm(3)), and with
the 128x128 kernel forced for every product (1)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import Model, _lib

dev = torch.device("cuda:0")
lib = _lib.load()
res = {}
for dim, depth in ((128, 6), (512, 12)):
    torch.manual_seed(0)
    m = Model(dim=dim, depth=depth, precision="hybrid").to(dev).eval()
    for B, N in ((1, 1024), (2, 1024), (4, 1024), (8, 1024), (16, 1024), (32, 1024)):
        x = torch.randn(B, N, dim, device=dev)
        t = torch.full((B,), 0.5, device=dev)
        for force in (0, 3, 1):
            lib.ns2_debug_force_gemm(force)
            with torch.no_grad():
                for _ in range(5):
                    m(x, t)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    m(x, t)
                torch.cuda.synchronize()
            res[f"d{dim}_b{B}/{ {0: 'auto_splitk', 3: 'no_split', 1: 'gemm128'}[force] }"] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
        lib.ns2_debug_force_gemm(0)
print(json.dumps(res, indent=1))
