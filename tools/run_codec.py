#!/usr/bin/env python
"""EnCodec front on the HIP path (SEANet encoder -> RVQ, SEANet decoder) a few times, for rocprofv3 --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transformers as tf
from naturalspeech2_pytorch_amd import EncodecWrapperHIP

dev = torch.device("cuda:0")
torch.manual_seed(0)
hf = tf.EncodecModel(tf.EncodecConfig()).eval()
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for layer in hf.quantizer.layers:
        layer.codebook.embed.copy_(torch.randn(layer.codebook.embed.shape, generator=g))
codec = EncodecWrapperHIP.from_hf(hf.to(dev), num_quantizers=8).to(dev)
wav = torch.randn(8, 1024 * 320, generator=g).to(dev)
with torch.no_grad():
    for _ in range(3):
        emb, codes, _ = codec(wav)
        rec = codec.decode(emb)
torch.cuda.synchronize()
print("ok", tuple(codes.shape), tuple(rec.shape))
