import sys, json, torch
sys.path.insert(0, '.')
import bench
r = bench.train_step_side(torch.device("cuda:0"))
for k, t in r.items():
    print(k, {kk: t[kk] for kk in ("ms_per_step", "train_precision", "mixed_ms_per_step", "exact_ms_per_step", "pytorch_composite_ms_per_step")})
