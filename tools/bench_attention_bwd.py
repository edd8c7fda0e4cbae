#!/usr/bin/env python
"""Attention forward (with lse) + backward of the training path at the headline shape (B=32, H=8, N=1024, d_head=64): ms per launch of
the forward, the delta pass and the two backward roles, algorithmic TFLOP/s (forward 4 B H N^2 64; backward 10 B H N^2 64: five
products), and the error of dq / dk / dv against torch autograd on one utterance."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naturalspeech2_pytorch_amd import ops, training

dev = torch.device("cuda:0")
B, H, N = 32, 8, 1024
a = H * 64
hb = training.HipBackend(3)
g = torch.Generator().manual_seed(3)
qkv = torch.randn(B * N, 3 * a, generator=g)
do = torch.randn(B * N, a, generator=g).to(dev)
p = hb.split(qkv.to(dev))
vt = hb.transpose(p, 2 * a, a, N, per_batch=True)
o, lse = hb.attention(p, 0, p, a, vt, B, H, N, N)
delta = hb.attention_delta(do, o, B, H, N)
do_row, _, _ = hb.grad_prep(do, a, want_row=True, attn=True)
dqkv = torch.empty(B * N, 3 * a, device=dev)


def timed(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


res = dict(
    forward_lse_ms=timed(lambda: hb.attention(p, 0, p, a, vt, B, H, N, N)),
    delta_ms=timed(lambda: hb.attention_delta(do, o, B, H, N)),
    dq_role_ms=timed(lambda: hb.attention_bwd(p, 0, p, a, p, 2 * a, do_row, lse, delta, B, H, N, N, dq=(dqkv, 0))),
    dkdv_role_ms=timed(lambda: hb.attention_bwd(p, 0, p, a, p, 2 * a, do_row, lse, delta, B, H, N, N, dkv=(dqkv, a, 2 * a))))
hb.attention_bwd(p, 0, p, a, p, 2 * a, do_row, lse, delta, B, H, N, N, dq=(dqkv, 0), dkv=(dqkv, a, 2 * a))
flop = B * H * N * N * 64.0
res["forward_tflops"] = 4 * flop / res["forward_lse_ms"] / 1e9
res["backward_ms"] = res["dq_role_ms"] + res["dkdv_role_ms"]
res["backward_tflops"] = 10 * flop / res["backward_ms"] / 1e9
res["backward_over_forward"] = res["backward_ms"] / res["forward_lse_ms"]
# utterance 0 against torch autograd on the plane values
pj = ops.join(p).cpu()[:N].double()
tq, tk, tv = (pj[:, i * a:(i + 1) * a].clone().requires_grad_(True) for i in range(3))
hd = lambda t: t.reshape(1, N, H, 64).transpose(1, 2)                     # noqa: E731
out = torch.nn.functional.scaled_dot_product_attention(hd(tq), hd(tk), hd(tv)).transpose(1, 2).reshape(N, a)
(out * ops.join(do_row).cpu()[:N].double()).sum().backward()
got = dqkv[:N].double().cpu()
res["rel_err_dq_dk_dv"] = [float(((got[:, i * a:(i + 1) * a] - t.grad).norm() / t.grad.norm())) for i, t in enumerate((tq, tk, tv))]
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))
