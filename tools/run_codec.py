#!/usr/bin/env python
"""EnCodec front on the HIP path (SEANet encoder -> RVQ, SEANet decoder) a few times, for rocprofv3 --stats / timing.
    python tools/run_codec.py [--batch 32] [--precision exact] [--decode]"""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transformers as tf
from naturalspeech2_pytorch_amd import EncodecWrapperHIP

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--precision", default="exact")
ap.add_argument("--decode", action="store_true")
ap.add_argument("--iters", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
hf = tf.EncodecModel(tf.EncodecConfig()).eval()
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for layer in hf.quantizer.layers:
        layer.codebook.embed.copy_(torch.randn(layer.codebook.embed.shape, generator=g))
codec = EncodecWrapperHIP.from_hf(hf.to(dev), num_quantizers=8, precision=args.precision).to(dev)
wav = torch.randn(args.batch, 1024 * 320, generator=g).to(dev)
with torch.no_grad():
    emb, codes, _ = codec(wav)                    # packs the weights
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        emb, codes, _ = codec(wav)
    torch.cuda.synchronize()
    t_enc = (time.perf_counter() - t0) / args.iters
    t_dec = None
    if args.decode:
        rec = codec.decode(emb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            rec = codec.decode(emb)
        torch.cuda.synchronize()
        t_dec = (time.perf_counter() - t0) / args.iters
print(f"ok codes {tuple(codes.shape)} precision {args.precision} batch {args.batch}: encode {t_enc * 1e3:.2f} ms"
      + (f", decode {t_dec * 1e3:.2f} ms" if t_dec else ""))
