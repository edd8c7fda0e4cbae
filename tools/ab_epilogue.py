"""A/B of epilogue builds at the headline shapes: prints one digest per op output (sha256 of the raw bytes), so two libraries
(NS2_LIB=...) can be compared bit for bit, and the GEGLU output's error against an fp64 torch evaluation on a row sample.
    python tools/ab_epilogue.py --prec 4 > a.txt ; NS2_LIB=.../libns2hip_g2_slowepi.so python tools/ab_epilogue.py --prec 4 > b.txt ; diff a.txt b.txt"""
import argparse, hashlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prec", type=int, default=4)
args = ap.parse_args()
P = args.prec
OP = 4 if P == 5 else P
dev = torch.device("cuda:0")
_lib.check(_lib.load().ns2_debug_force_gemm(2))
B, N, d, f = 32, 1024, 512, 1365
M = B * N
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
dig = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
fp = ops.round_up(f, 32)
X512, XF = rnd(M, d), rnd(M, f)
x512, xf = ops.split(X512, precision=OP), ops.split(XF, ldo=fp, precision=OP)
w = ops.PackedWeight(rnd(f, f, 3, scale=0.02), precision=OP); b = rnd(f)
print("ffconv ", dig(ops.linear_split(w, xf, bias=b, conv_taps=3, dilation=1, seq_len=N, precision=OP).buf))
W1 = rnd(2 * f, d, scale=0.04); B1 = rnd(2 * f)
w1 = ops.PackedWeight(W1, geglu=True, precision=OP); pb = ops.geglu_pack_bias(B1, f)
og = ops.linear_geglu(w1, x512, pb, precision=OP)
print("geglu  ", dig(og.buf))
rows = torch.arange(0, M, 257, device=dev)
h = X512[rows].double() @ W1.double().t() + B1.double()
ref = torch.nn.functional.gelu(h[:, f:]) * h[:, :f]
got = ops.join(og, f)[rows].double()
print(f"geglu rel err vs fp64 on {len(rows)} rows: {((got - ref).norm() / ref.norm()).item():.3e}  max abs {(got - ref).abs().max().item():.3e}")
w2 = ops.PackedWeight(rnd(d, f, scale=0.03), precision=OP); b2 = rnd(d); r = rnd(M, d)
print("ffout  ", dig(ops.linear_f32(w2, xf, bias=b2, resid=r, precision=OP)))
wq = ops.PackedWeight(rnd(1536, d, scale=0.04), precision=OP)
qk, vt = ops.linear_qkv(wq, x512, seq_len=N, split_col=1024, precision=OP)
print("qkv.qk ", dig(qk.buf)); print("qkv.vt ", dig(vt.buf))
wo = ops.PackedWeight(rnd(d, d, scale=0.04), precision=OP); ro = rnd(M, d)
print("outproj", dig(ops.linear_f32(wo, x512, resid=ro, precision=OP)))
ww = ops.PackedWeight(rnd(d, d, 3, scale=0.03), extra1x1=rnd(d, d, 1, scale=0.04), precision=OP); bc, br = rnd(d), rnd(d); film = rnd(B, 2 * d)
print("wavenet", dig(ops.wavenet_block(ww, x512, N, 16, bc, br, film, precision=P).buf))
print("saturation count", ops.saturation_count(reset=True))
