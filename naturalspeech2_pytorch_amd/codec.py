"""RVQ side of the codec boundary (`EncodecWrapper`, SURVEY §8b "Codec surface") with the residual-VQ encode and
decode running in the HIP kernels of csrc/rvq.hip.

Surface kept for `NaturalSpeech2` (reference call sites in brackets):
    codec.target_sample_hz, .seq_len_multiple_of [NS2:1213-1214], .codebook_dim [NS2:1244], .eval() [NS2:1444, 1610]
    codec(x, return_encoded=True, curtail_from_left=...) -> (emb [b,n,128], codes [b,n,Q] int64, None)   [NS2:1445, 1611]
    codec.decode(emb [b,n,128]) -> waveform [b,1,T]                                                        [NS2:1496-1499]
    codec.rq(x_start, codes) -> (quantized, ce_loss)                                                       [NS2:1682]

The SEANet conv/LSTM encoder and decoder of EnCodec are callables (`encoder`, `decoder`): `from_hf` takes them from a HF
`transformers.EncodecModel` (the in-container stand-in for the un-vendored `encodec` package), by default wrapped
in `seanet.SEANetEncoderHIP / SEANetDecoderHIP` so that waveform <-> latents also runs on the HIP kernels, and
copies its codebooks.  With `encoder=None` the wrapper accepts latents [b, n, 128] directly.
"""
from typing import Callable, Optional

import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from ._cache import PackedCache


class HipRVQ(nn.Module):
    """EnCodec residual vector quantizer (HFENC:424-447) over `codebooks` [Q, C, 128]."""

    def __init__(self, codebooks: torch.Tensor, tie_eps: float = 1e-4):
        super().__init__()
        assert codebooks.ndim == 3 and codebooks.shape[-1] == 128 and codebooks.shape[1] % 64 == 0
        self.register_buffer("codebooks", codebooks.float().contiguous())
        self.tie_eps = tie_eps
        self._norm = PackedCache()

    def _cb_norm(self):
        return self._norm.get([self.codebooks], lambda: ops.rvq_prepare(self.codebooks))

    @torch.no_grad()
    def encode(self, latents: torch.Tensor):
        """latents [b, n, 128] -> codes [b, n, Q] int64, emb [b, n, 128] (sum of the selected code vectors)."""
        b, n, d = latents.shape
        x = latents.reshape(b * n, d).float().contiguous()
        codes, emb = ops.rvq_encode(x, self.codebooks, self._cb_norm(), tie_eps=self.tie_eps)
        return codes.reshape(b, n, -1), emb.reshape(b, n, d)

    @torch.no_grad()
    def decode(self, codes: torch.Tensor):
        b, n, q = codes.shape
        return ops.rvq_decode(codes.reshape(b * n, q).contiguous(), self.codebooks).reshape(b, n, -1)


class ResidualVQCrossEntropy(nn.Module):
    """`codec.rq(x, indices) -> (quantized_out, ce_loss)`: the training-time cross-entropy of the diffusion model's predicted
    x_start against the codec's own code indices (NS2:1670-1684, off by default: rvq_cross_entropy_loss_weight = 0).

    PARITY UNPINNED: upstream this is `vector_quantize_pytorch.ResidualVQ.forward(x, indices=...)` (setup.py pins only
    `>=1.4.1`, no lock file; not vendored, not installed here).  Restated from its published algorithm: per quantizer the
    logits are the NEGATIVE EUCLIDEAN distances (with the square root) of the running residual to every code, the loss is
    `F.cross_entropy(logits, indices[..., q])`, the residual is reduced by the nearest code (detached), and the layer losses
    are summed.  Differentiable in `x` (autograd composite: the term only exists in training)."""

    def __init__(self, owner: HipRVQ):
        super().__init__()
        self._owner = [owner]              # not registered: the codebooks stay owned (and moved) by the HipRVQ

    def forward(self, x, indices):
        cb = self._owner[0].codebooks.to(x.dtype)
        assert indices.shape[:-1] == x.shape[:-1] and indices.shape[-1] == cb.shape[0]
        assert not torch.any(indices == -1), "some of the residual vq indices were dropped out"
        residual, out, ce = x, torch.zeros_like(x), 0.
        for q in range(cb.shape[0]):
            e = cb[q]
            d2 = residual.pow(2).sum(-1, keepdim=True) - 2 * residual @ e.t() + e.pow(2).sum(-1)
            logits = -d2.clamp(min=0).sqrt()                                     # [b, n, C]
            ce = ce + F.cross_entropy(logits.transpose(1, 2), indices[..., q], ignore_index=-1)
            quantized = F.embedding(logits.argmax(dim=-1), e)
            residual = residual - quantized.detach()
            out = out + quantized
        return out, ce


class EncodecWrapperHIP(nn.Module):
    target_sample_hz = 24000
    seq_len_multiple_of = 320          # strides 2*4*5*8
    codebook_dim = 128

    def __init__(self, codebooks: torch.Tensor, encoder: Optional[Callable] = None, decoder: Optional[Callable] = None):
        super().__init__()
        self._hip_codec_init(codebooks, encoder, decoder)

    def _hip_codec_init(self, codebooks, encoder=None, decoder=None):
        """everything but nn.Module's own set-up: compat.hip_backed_codec_class builds a subclass of the REFERENCE's codec class whose
        own __init__ (it fetches pretrained EnCodec weights) must not run"""
        self.rvq = HipRVQ(codebooks)
        self.rq = ResidualVQCrossEntropy(self.rvq)
        self.encoder, self.decoder = encoder, decoder

    @classmethod
    def from_hf(cls, hf_model, num_quantizers: int = 8, hip_seanet: bool = True, precision: str = "exact"):
        """wire a `transformers.EncodecModel`'s SEANet encoder / decoder and its first `num_quantizers` codebooks (6 kbps at
        24 kHz = 8, what audiolm's EncodecWrapper uses).  hip_seanet=True (default): the SEANet stacks run on the HIP kernels
        (seanet.py) with the model's own (weight-normalised) parameters; False: HF's PyTorch modules are called as they are."""
        layers = list(hf_model.quantizer.layers)[:num_quantizers]
        cbs = torch.stack([l.codebook.embed.detach().float() for l in layers])
        enc, dec = hf_model.encoder, hf_model.decoder
        if hip_seanet:
            from .seanet import SEANetDecoderHIP, SEANetEncoderHIP
            enc, dec = SEANetEncoderHIP(enc, precision=precision), SEANetDecoderHIP(dec, precision=precision)
        return cls(cbs, encoder=enc, decoder=dec)

    @property
    def num_quantizers(self):
        return self.rvq.codebooks.shape[0]

    def refresh_weights(self):
        """run-boundary content check of everything this wrapper packed (SEANet stacks, codebook norms): an EMA copy of the whole
        `NaturalSpeech2` (NS2:1793) rewrites these through `.data`"""
        for m in (self.encoder, self.decoder):
            if hasattr(m, "refresh_weights"):
                m.refresh_weights()
        self.rvq._norm.refresh([self.rvq.codebooks])

    @torch.no_grad()
    def forward(self, x, return_encoded=True, curtail_from_left=False, **kwargs):
        """x: raw audio [b, t] (needs `encoder`) or latents [b, n, 128] -> (emb [b,n,128], codes [b,n,Q], None)."""
        if x.ndim == 2:
            if self.encoder is None:
                raise RuntimeError("raw audio needs the SEANet encoder (out of scope of the HIP path): pass encoder=")
            t = x.shape[-1] // self.seq_len_multiple_of * self.seq_len_multiple_of
            if t == 0:
                raise ValueError(f"audio shorter than one codec frame ({self.seq_len_multiple_of} samples)")   # x[..., -0:] would keep everything
            x = x[..., -t:] if curtail_from_left else x[..., :t]
            latents = self.encoder(x[:, None]).transpose(1, 2)       # [b, 128, n] -> [b, n, 128]
        else:
            latents = x
        codes, emb = self.rvq.encode(latents)
        return emb, codes, None

    @torch.no_grad()
    def decode(self, emb):
        if self.decoder is None:
            raise RuntimeError("waveform decode needs the SEANet decoder (out of scope of the HIP path): pass decoder=")
        return self.decoder(emb.transpose(1, 2))
