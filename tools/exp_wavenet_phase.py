"""What would the Wavenet block's dilated-conv phase cost on DENSE half operands?  Times, at the headline shape (M = 32768, d = 512):
(a) ns2_wavenet_block at precision 5 (the benched kernel: phase 1 gathers the half parts of FMT_H8 lines = two partial lines per row
and 64-deep tile), (b) the same dilated conv alone as a precision-2 EPI_SPLIT GEMM on dense IEEE-half planes (one full line per row
and tile), (c) res_conv alone as a precision-4 GEMM on FMT_H8 lines."""
import os, sys, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import ops

dev = torch.device("cuda:0")
B, N, d = 32, 1024, 512
M = B * N
g = torch.Generator().manual_seed(0)
x = torch.randn(M, d, generator=g).to(dev)
wc = (torch.randn(d, d, 3, generator=g) * 0.03).to(dev)
wr = (torch.randn(d, d, 1, generator=g) * 0.03).to(dev)
bc, br = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
film = torch.cat((torch.ones(B, d), torch.zeros(B, d)), 1).to(dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {}
x4 = ops.split(x, precision=4)
x2 = ops.split(x, precision=2)
w_blk = ops.PackedWeight(wc, extra1x1=wr, precision=4)
w_c2 = ops.PackedWeight(wc, precision=2)
w_c4 = ops.PackedWeight(wc, precision=4)
w_r4 = ops.PackedWeight(wr[:, :, 0].contiguous(), precision=4)
for dil in (1, 8, 128):
    res[f"dil{dil}"] = dict(
        block_p5_ms=timed(lambda: ops.wavenet_block(w_blk, x4, N, dil, bc, br, film, precision=5)),
        block_p4_ms=timed(lambda: ops.wavenet_block(w_blk, x4, N, dil, bc, br, film, precision=4)),
        conv_dense_half_ms=timed(lambda: ops.linear_split(w_c2, x2, bias=bc, conv_taps=3, dilation=dil, seq_len=N, precision=2)),
        conv_mixed_ms=timed(lambda: ops.linear_split(w_c4, x4, bias=bc, conv_taps=3, dilation=dil, seq_len=N, precision=4)),
        res_mixed_ms=timed(lambda: ops.linear_split(w_r4, x4, bias=br, precision=4)))
print(json.dumps(res, indent=1))
