"""Block-level timeline of the 256x256 GEMM family from a -DG2_BLKTRACE build (tools/ablate_gemm2.sh blk -DG2_BLKTRACE):
    NS2_LIB=naturalspeech2_pytorch_amd/libns2hip_g2_blk.so python tools/trace_blocks.py [--prec 4] [--which qkv,ffin,...]
Per block and wave the kernel stamps the 100 MHz s_memrealtime at entry / first K tile landed / K loop done / epilogue stores
issued / stores drained, and HW_ID + XCC_ID.  Reported: where a block's time goes, and how a CU's consecutive blocks follow
each other (launch -> first block start, gaps between blocks on one CU, tail)."""
import argparse, ctypes, os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naturalspeech2_pytorch_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--prec", type=int, default=4)
ap.add_argument("--which", default="qkv,ffin,outproj,ffout,ffconv,wavenet")
ap.add_argument("--out", default="")
args = ap.parse_args()
lib = _lib.load(); _lib.check(lib.ns2_debug_force_gemm(2))
raw = ctypes.CDLL(os.environ["NS2_LIB"])
dev = torch.device("cuda:0")
B, N, d, f = 32, 1024, 512, 1365
M = B * N
g = torch.Generator().manual_seed(0)
P = args.prec
OP = 4 if P == 5 else P          # operand format of the planes / weights (5 = hybrid wavenet block on precision-4 operands)


def rnd(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(dev)


fp = ops.round_up(f, 32)
x512 = ops.split(rnd(M, d), precision=OP)
xf = ops.split(rnd(M, f), ldo=fp, precision=OP)
cases = {}
w = ops.PackedWeight(rnd(f, f, 3, scale=0.02), precision=OP); b = rnd(f)
cases["ffconv"] = (lambda: ops.linear_split(w, xf, bias=b, conv_taps=3, dilation=1, seq_len=N, precision=OP), 6 * 128, 2.0 * M * f * 3 * f)
w1 = ops.PackedWeight(rnd(2 * f, d, scale=0.04), geglu=True, precision=OP); pb = ops.geglu_pack_bias(rnd(2 * f), f)
cases["ffin"] = (lambda: ops.linear_geglu(w1, x512, pb, precision=OP), 11 * 128, 2.0 * M * d * 2 * f)
w2 = ops.PackedWeight(rnd(d, f, scale=0.03), precision=OP); b2 = rnd(d); r = rnd(M, d)
cases["ffout"] = (lambda: ops.linear_f32(w2, xf, bias=b2, resid=r, precision=OP), 2 * 128, 2.0 * M * f * d)
wq = ops.PackedWeight(rnd(1536, d, scale=0.04), precision=OP)
cases["qkv"] = (lambda: ops.linear_qkv(wq, x512, seq_len=N, split_col=1024, precision=OP), 6 * 128, 2.0 * M * d * 1536)
wo = ops.PackedWeight(rnd(d, d, scale=0.04), precision=OP); ro = rnd(M, d)
cases["outproj"] = (lambda: ops.linear_f32(wo, x512, resid=ro, precision=OP), 2 * 128, 2.0 * M * d * d)
ww = ops.PackedWeight(rnd(d, d, 3, scale=0.03), extra1x1=rnd(d, d, 1, scale=0.04), precision=OP); bc, br = rnd(d), rnd(d); film = rnd(B, 2 * d)
cases["wavenet"] = (lambda: ops.wavenet_block(ww, x512, N, 16, bc, br, film, precision=P), 2 * 128, 2.0 * M * d * 4 * d)

report = {}
for name in args.which.split(","):
    fn, nblk, flops = cases[name]
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    buf = (ctypes.c_ulonglong * (64 * nblk))()
    assert raw.ns2_debug_read_blocks(buf, nblk) == 0
    a = np.array(buf, dtype=np.int64).reshape(nblk, 8, 8)
    t = a[:, :, :5].astype(np.float64) / 100.0            # us
    t0 = t[:, :, 0].min()
    hw = a[:, 0, 5]
    cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf)
    active = a[:, :, 7] > 0
    cyc = a[:, :, 6].astype(np.float64)
    dur = t[:, :, 4] - t[:, :, 0]
    clk = (cyc / np.maximum(dur, 1e-3)).mean()
    seg = np.diff(t, axis=2)                              # [blk][wave][prologue, kloop, epi issue, drain]
    act = active
    m = lambda k: float(seg[:, :, k][act].mean())
    blk_start = t[:, :, 0].min(axis=1); blk_end = t[:, :, 4].max(axis=1)
    # per-CU succession
    gaps, per_cu = [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        o = idx[np.argsort(blk_start[idx])]
        per_cu.append(len(o))
        for i in range(1, len(o)):
            gaps.append(blk_start[o[i]] - blk_end[o[i - 1]])
    gaps = np.array(gaps) if gaps else np.zeros(1)
    rep = dict(launch_us=us, tflops=flops / us / 1e6, nblk=nblk, cus=int(len(np.unique(cu))), blocks_per_cu_max=int(max(per_cu)),
               shader_mhz=clk, block_us=float((blk_end - blk_start).mean()), prologue_us=m(0), kloop_us=m(1), epi_issue_us=m(2),
               drain_us=m(3), first_start_to_last_end_us=float(blk_end.max() - t0), gap_mean_us=float(gaps.mean()),
               gap_min_us=float(gaps.min()), gap_max_us=float(gaps.max()), overlap_frac=float((gaps < 0).mean()),
               first_block_start_spread_us=float(np.percentile(blk_start - t0, 95)))
    report[name] = rep
    print(f"{name:8s} prec={P}: launch {us:7.1f} us ({rep['tflops']:6.0f} TF)  blocks {nblk} on {rep['cus']} CUs (max {rep['blocks_per_cu_max']}/CU)  "
          f"clk {clk:5.0f} MHz | block {rep['block_us']:6.1f} us = prologue {rep['prologue_us']:5.2f} + K loop {rep['kloop_us']:6.2f} + epilogue issue "
          f"{rep['epi_issue_us']:5.2f} + drain {rep['drain_us']:5.2f} | CU gap between blocks mean {rep['gap_mean_us']:5.2f} (min {rep['gap_min_us']:.2f}, "
          f"max {rep['gap_max_us']:.2f}, overlapped {rep['overlap_frac']:.2f}) | span {rep['first_start_to_last_end_us']:.1f} us")
if args.out:
    json.dump(report, open(args.out, "w"), indent=1)
