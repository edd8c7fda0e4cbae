"""RVQ side of the codec boundary (`EncodecWrapper`, SURVEY §8b "Codec surface") with the residual-VQ encode and
decode running in the HIP kernels of csrc/rvq.hip.

The SEANet conv/LSTM encoder and decoder of EnCodec are out of scope (they stay on PyTorch-ROCm /
MIOpen); they are injected as callables so that `codec(x, return_encoded=True)`, `codec.decode(emb)` and the
attributes `NaturalSpeech2.__init__` reads (`target_sample_hz`, `seq_len_multiple_of`, `codebook_dim`;
NS2:1213-1214, 1244) keep working.  With `encoder=None` the wrapper accepts latents [b, n, 128] directly.
"""
from typing import Callable, Optional

import torch
from torch import nn

from . import ops


class HipRVQ(nn.Module):
    """EnCodec residual vector quantizer (HFENC:424-447) over `codebooks` [Q, C, 128]."""

    def __init__(self, codebooks: torch.Tensor, tie_eps: float = 1e-4):
        super().__init__()
        assert codebooks.ndim == 3 and codebooks.shape[-1] == 128 and codebooks.shape[1] % 64 == 0
        self.register_buffer("codebooks", codebooks.float().contiguous())
        self.tie_eps = tie_eps
        self._norm = None
        self._norm_sig = None

    def _cb_norm(self):
        sig = (self.codebooks.data_ptr(), self.codebooks._version)
        if self._norm is None or self._norm_sig != sig:
            self._norm = ops.rvq_prepare(self.codebooks)
            self._norm_sig = sig
        return self._norm

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


class EncodecWrapperHIP(nn.Module):
    target_sample_hz = 24000
    seq_len_multiple_of = 320          # strides 2*4*5*8
    codebook_dim = 128

    def __init__(self, codebooks: torch.Tensor, encoder: Optional[Callable] = None, decoder: Optional[Callable] = None):
        super().__init__()
        self.rvq = HipRVQ(codebooks)
        self.encoder, self.decoder = encoder, decoder

    @property
    def num_quantizers(self):
        return self.rvq.codebooks.shape[0]

    @torch.no_grad()
    def forward(self, x, return_encoded=True, curtail_from_left=False, **kwargs):
        """x: raw audio [b, t] (needs `encoder`) or latents [b, n, 128] -> (emb [b,n,128], codes [b,n,Q], None)."""
        if x.ndim == 2:
            if self.encoder is None:
                raise RuntimeError("raw audio needs the SEANet encoder (out of scope of the HIP path): pass encoder=")
            t = x.shape[-1] // self.seq_len_multiple_of * self.seq_len_multiple_of
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
