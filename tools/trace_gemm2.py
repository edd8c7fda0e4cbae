"""Phase timeline of the 256x256 GEMM (FF conv shape) from a -DG2_TRACE build:  tools/ablate_gemm2.sh trace -DG2_TRACE ;
NS2_LIB=.../libns2hip_g2_trace.so python tools/trace_gemm2.py [--prec 3]"""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naturalspeech2_pytorch_amd import _lib, ops

ap = argparse.ArgumentParser(); ap.add_argument("--prec", type=int, default=3); args = ap.parse_args()
lib = _lib.load(); _lib.check(lib.ns2_debug_force_gemm(2))
raw = ctypes.CDLL(os.environ["NS2_LIB"])
g = torch.Generator().manual_seed(0)
M, f, N = 32768, 1365, 1024
x = ops.split((torch.randn(M, f, generator=g)).cuda(), ldo=ops.round_up(f, 32), precision=args.prec)
w = ops.PackedWeight((torch.randn(f, f, 3, generator=g) * 0.02).cuda(), precision=args.prec); b = torch.randn(f, generator=g).cuda()
for _ in range(5):
    ops.linear_split(w, x, bias=b, conv_taps=3, dilation=1, seq_len=N, precision=args.prec)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 64 * 10))()
assert raw.ns2_debug_read_trace(buf) == 0
full = np.array(buf, dtype=np.int64).reshape(8, 64, 10)
t = full[:, :, :9]
nt = min(64, (3 * ops.round_up(f, 64 if args.prec in (1, 2) else 32)) // (64 if args.prec in (1, 2) else 32) - 16)      # tiles actually traced
rt = full[0, :nt, 9]
cyc = full[0, :nt, 0]
print(f"shader clock over the traced tiles: {(cyc[-1] - cyc[4]) / ((rt[-1] - rt[4]) / 100.0):.0f} MHz "
      f"({nt} tiles, {(cyc[-1] - cyc[4])} cycles in {(rt[-1] - rt[4]) / 100.0:.2f} us)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.linear_split(w, x, bias=b, conv_taps=3, dilation=1, seq_len=N, precision=args.prec)
e1.record(); torch.cuda.synchronize()
print(f"traced build: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch")
# stamps: tile start | early DMA issued (waves 0-3) | K-step-0 fragments landed | K-step-0 MFMAs issued | late DMA issued
# (waves 4-7) | K-step-1 fragments landed | K-step-1 MFMAs issued [fast mode: steps 2,3 follow] | own DMA landed | barrier passed
names = ["dma_early", "rd0", "mfma0", "dma_late", "rd1", "mfma1", "vmcnt", "barrier"]
d = np.diff(t, axis=2).astype(np.float64)                     # [wave][tile][8 phases]
tile = (t[:, 1:nt, 0] - t[:, :nt - 1, 0]).astype(np.float64)
print("cycles per tile (wave mean):", np.round(tile[:, 4:].mean(axis=1), 0))
print("phase means per wave (cycles):   " + "  ".join(f"{n:>9s}" for n in names))
for wv in range(8):
    print(f"  wave {wv}:                        " + "  ".join(f"{v:9.0f}" for v in d[wv, 4:nt].mean(axis=0)))
