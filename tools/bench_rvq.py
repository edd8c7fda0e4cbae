"""BASELINE config 4: EnCodec RVQ encode of 32 x 1024 frames x 8 codebooks of 1024 x 128 (fp32), bit-exact indices.
    python tools/bench_rvq.py [--iters 20] [--frames 1024] [--batch 32]
Prints one JSON line: HIP kernel time, fp32-MFMA roofline fraction (157.3 TF), CPU oracle time on a bounded sample."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import ops  # noqa: E402
from oracle import rvq_oracle as R  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--frames", type=int, default=1024)
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, Q, C, D = args.batch * args.frames, 8, 1024, 128
x = torch.randn(M, D, generator=g)
cb = torch.randn(Q, C, D, generator=torch.Generator().manual_seed(1))
xd, cbd = x.to(dev), cb.to(dev)
norm = ops.rvq_prepare(cbd)
codes, emb, ties = ops.rvq_encode(xd, cbd, norm, count_ties=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.iters):
    ops.rvq_encode(xd, cbd, norm)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.iters
flops = 2.0 * M * D * C * Q
# parity on a bounded sample (first 4096 frames) against the oracle, timed as the CPU baseline
n = min(M, 4096)
t0 = time.perf_counter()
c_ref, e_ref, _ = R.rvq_encode(x[:n], cb)
cpu_s = time.perf_counter() - t0
mism = int((codes[:n].cpu() != c_ref).sum())
print(json.dumps({
    "metric": "EnCodec RVQ encode, %d x %d frames x 8 codebooks (1024 x 128) fp32" % (args.batch, args.frames),
    "ms": round(ms, 4), "frames_per_s": round(M / ms * 1e3), "algorithmic_gflop": round(flops / 1e9, 2),
    "roofline": {"bound": "mfma (fp32 v_mfma_f32_32x32x2_f32)", "achieved": round(flops / ms / 1e9, 2), "peak": 157.3,
                 "unit": "TFLOP/s", "frac": round(flops / ms / 1e9 / 157.3, 4)},
    "near_tie_rechecks": int(ties.item()), "index_mismatches_vs_oracle_first_%d" % n: mism,
    "emb_bit_exact": bool(torch.equal(emb[:n].cpu(), e_ref)),
    "cpu_baseline": {"kind": "port", "value_frames_per_s": round(n / cpu_s), "cores": torch.get_num_threads(),
                     "sample": "oracle rvq_encode on %d frames: %.3f s" % (n, cpu_s)},
}))
