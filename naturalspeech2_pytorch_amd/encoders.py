"""Conditioning encoders built from the plain `Transformer` (the first "next" row): `SpeechPromptEncoder`
(NS2:289-341: 8 x [Conv1d k=9 "same" + SiLU] -> Transformer) and `PhonemeEncoder` (NS2:228-287: Embedding ->
CausalConv1d k=9 + SiLU -> Transformer).  They run once per utterance, not per denoising step; they reuse the hot path's
kernels: the k=9 convolutions are 9-tap shifted-row GEMMs with the SiLU in the epilogue (csrc/gemm*.hip, pad_left/act).
Same constructor keywords and state_dict keys as the reference classes; string input (tokenizer / espeak) is out of scope.
"""
from typing import Tuple

import torch
from torch import nn

from . import ops
from ._cache import PackedCache
from .model import _NoParams, _PRECISIONS
from .transformer import Transformer, needs_autograd


class _ConvStack(PackedCache):
    """packed k-tap convs, re-packed when the parameters change (by content, checked on every call: these encoders run once per
    utterance); dropped by deepcopy / pickling."""

    def __init__(self, fingerprint_every=1):
        super().__init__(fingerprint_every)

    def packed_for(self, convs, prec):
        ts = [t for c in convs for t in (c.weight, c.bias)]
        return self.get(ts, lambda: [(ops.PackedWeight(c.weight.detach().float().contiguous(), precision=prec),
                                      c.bias.detach().float().contiguous()) for c in convs], extra=(prec,))


class SpeechPromptEncoder(nn.Module):
    def __init__(self, dim_codebook, dims: Tuple[int, ...] = (256, 2048, 2048, 2048, 2048, 512, 512, 512), *, depth=6, heads=8,
                 dim_head=64, dropout=0.2, kernel_size=9, padding=4, use_flash_attn=True, precision="exact"):
        super().__init__()
        dims = [dim_codebook, *dims]
        self.dim, self.dim_out = dims[0], dims[-1]
        assert precision in _PRECISIONS, f"precision must be one of {sorted(_PRECISIONS)}"
        self.kernel_size, self.padding, self.precision = kernel_size, padding, precision
        mods = [_NoParams()]
        for d_in, d_out in zip(dims[:-1], dims[1:]):
            mods.extend([nn.Conv1d(d_in, d_out, kernel_size, padding=padding), _NoParams()])
        mods.append(_NoParams())
        self.conv = nn.Sequential(*mods)                      # same indices as the reference Sequential (Rearrange, conv, SiLU, ...)
        self.transformer = Transformer(dim=dims[-1], depth=depth, heads=heads, dim_head=dim_head, dropout=dropout,
                                       use_flash=use_flash_attn, precision=precision)
        self._stack = _ConvStack()

    def forward(self, x):
        """Under autograd (the reference trains prompt_enc jointly, NS2:1542-1543) the differentiable composite runs; inference
        runs in the HIP kernels."""
        assert x.shape[-1] == self.dim
        if needs_autograd(self, x):
            from .autograd_path import speech_prompt_encoder_autograd
            return speech_prompt_encoder_autograd(self, x)
        return self._forward_hip(x)

    def refresh_weights(self):
        convs = [m for m in self.conv if isinstance(m, nn.Conv1d)]
        self._stack.refresh([t for c in convs for t in (c.weight, c.bias)])
        self.transformer.refresh_weights()

    @torch.no_grad()
    def _forward_hip(self, x):
        b, n, _ = x.shape
        prec = _PRECISIONS[self.precision]
        convs = [m for m in self.conv if isinstance(m, nn.Conv1d)]
        packed = self._stack.packed_for(convs, prec)
        h = ops.split(x.reshape(b * n, self.dim).float().contiguous(), precision=prec)
        for i, (pw, bias) in enumerate(packed):
            kw = dict(bias=bias, conv_taps=self.kernel_size, dilation=1, seq_len=n, pad_left=self.padding, act=1, precision=prec)
            if i + 1 < len(packed):
                h = ops.linear_split(pw, h, **kw)
            else:
                h = ops.linear_f32(pw, h, **kw)
        return self.transformer._forward_hip(h.reshape(b, n, self.dim_out)).to(x.dtype)


class PhonemeEncoder(nn.Module):
    def __init__(self, *, tokenizer=None, num_tokens=None, dim=512, dim_hidden=512, kernel_size=9, depth=6, dim_head=64, heads=8,
                 conv_dropout=0.2, attn_dropout=0., use_flash=False, precision="exact"):
        super().__init__()
        if tokenizer is not None and num_tokens is None:
            num_tokens = tokenizer.vocab_size
        assert num_tokens is not None, "token ids are required (the text front-end is out of scope)"
        self.tokenizer = tokenizer
        self.token_emb = nn.Embedding(num_tokens + 1, dim)
        self.pad_id = num_tokens
        assert precision in _PRECISIONS, f"precision must be one of {sorted(_PRECISIONS)}"
        self.kernel_size, self.dim_hidden, self.precision, self.conv_dropout = kernel_size, dim_hidden, precision, conv_dropout
        self.conv = nn.Sequential(_NoParams(), nn.Conv1d(dim, dim_hidden, kernel_size), _NoParams(), _NoParams(), _NoParams())
        self.transformer = Transformer(dim=dim_hidden, depth=depth, dim_head=dim_head, heads=heads, dropout=attn_dropout,
                                       use_flash=use_flash, precision=precision)
        self._stack = _ConvStack()

    def forward(self, x, mask=None):
        if not torch.is_tensor(x):
            raise NotImplementedError("List[str] input needs the tokenizer / espeak front-end (out of scope); pass token ids")
        if needs_autograd(self):
            from .autograd_path import phoneme_encoder_autograd
            return phoneme_encoder_autograd(self, x, mask)
        return self._forward_hip(x, mask)

    def refresh_weights(self):
        self._stack.refresh([self.conv[1].weight, self.conv[1].bias])
        self.transformer.refresh_weights()

    @torch.no_grad()
    def _forward_hip(self, x, mask=None):
        b, n = x.shape
        prec = _PRECISIONS[self.precision]
        emb = ops.embedding(x, self.token_emb.weight.detach().float().contiguous(), self.pad_id)     # NS2:281-284
        (pw, bias), = self._stack.packed_for([self.conv[1]], prec)
        h = ops.linear_f32(pw, ops.split(emb.reshape(b * n, -1), precision=prec), bias=bias, conv_taps=self.kernel_size, dilation=1, seq_len=n,
                           pad_left=-1, act=1, precision=prec)                                        # CausalConv1d + SiLU
        return self.transformer._forward_hip(h.reshape(b, n, self.dim_hidden), mask=mask)
