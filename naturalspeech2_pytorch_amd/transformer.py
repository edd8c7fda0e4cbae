"""`Transformer` — the reference's plain pre-norm encoder (NS2:1073-1115: RMSNorm -> Attention -> RMSNorm -> FeedForward
without conv, optional key-padding mask), the block `PhonemeEncoder` / `SpeechPromptEncoder` are built from.  It is not on
the per-step path; it shares the hot path's HIP kernels and is composed here from the op-level C ABI.
Same constructor keywords and state_dict keys as the reference class.
"""
import torch
from torch import nn

from . import ops
from ._cache import PackedCache
from .model import _Attention, _RMSNorm, _feedforward, _PRECISIONS


_warned_eval_no_graph = False


def needs_autograd(module, x=None):
    """Which path a per-utterance module (plain `Transformer`, `SpeechPromptEncoder`, `PhonemeEncoder`) takes: the differentiable
    PyTorch composite when a gradient can be wanted -- grad mode on AND (the input carries one, or the module is in training mode
    with trainable parameters) -- else the forward-only HIP kernels.  An `eval()` module called without `torch.no_grad()` keeps
    the HIP kernels and its `precision` (a freshly built module has requires_grad parameters: ADVICE r3) -- but PyTorch's `eval()`
    does not switch autograd off, so that call records NO graph where the reference would: it warns once, and
    `module.force_autograd = True` (or `.train()`, or an input that requires grad) selects the composite."""
    if not torch.is_grad_enabled():
        return False
    if (x is not None and torch.is_tensor(x) and x.requires_grad) or getattr(module, "force_autograd", False):
        return True
    trainable = any(p.requires_grad for p in module.parameters())
    if module.training and trainable:
        return True
    if trainable:
        global _warned_eval_no_graph
        if not _warned_eval_no_graph:
            _warned_eval_no_graph = True
            import warnings
            warnings.warn(f"{type(module).__name__} is in eval() mode with grad enabled and trainable parameters: running the forward-only "
                          f"HIP kernels, no autograd graph is recorded.  For gradients call .train(), set .force_autograd = True, or "
                          f"pass an input that requires grad; for inference wrap the call in torch.no_grad().")
    return False


class Transformer(nn.Module):
    def __init__(self, dim, *, depth, causal=False, dim_head=64, heads=8, use_flash=False, dropout=0., ff_mult=4,
                 final_norm=False, precision="exact"):
        super().__init__()
        assert dim_head in (32, 64, 128), "the HIP attention kernel is built for head dims 32, 64 (the reference default) and 128"
        self.dim, self.depth, self.heads, self.causal, self.dropout, self.dim_head = dim, depth, heads, causal, dropout, dim_head
        assert precision in _PRECISIONS, f"precision must be one of {sorted(_PRECISIONS)}"
        self.precision = precision
        self.layers = nn.ModuleList([
            nn.ModuleList([_RMSNorm(dim), _Attention(dim, dim_head, heads), _RMSNorm(dim), _feedforward(dim, ff_mult, False)])
            for _ in range(depth)])
        self.norm = _RMSNorm(dim) if final_norm else nn.Identity()
        self._cache = PackedCache(fingerprint_every=1)         # once per utterance: check the content every call (EMA copies)

    def refresh_weights(self):
        """re-pack iff the parameter CONTENTS changed since they were packed (run boundaries: NaturalSpeech2.sample / .forward)"""
        self._cache.refresh(self.parameters())

    def _pack(self):
        return self._cache.get(self.parameters(), self._build_packed, extra=(self.precision,))

    def _build_packed(self):
        prec = _PRECISIONS[self.precision]
        packed = []
        for norm1, attn, norm2, ff in self.layers:
            wqkv = torch.cat((attn.to_q.weight, attn.to_kv.weight), dim=0).detach().float().contiguous()
            w1, w2 = getattr(ff, "0"), getattr(ff, "2")
            f = w1.weight.shape[0] // 2
            packed.append(dict(
                g1=norm1.gamma.detach().float().contiguous(), g2=norm2.gamma.detach().float().contiguous(),
                qkv=ops.PackedWeight(wqkv, precision=prec),
                out=ops.PackedWeight(attn.to_out.weight.detach().float().contiguous(), precision=prec),
                w1=ops.PackedWeight(w1.weight.detach().float().contiguous(), geglu=True, precision=prec),
                b1=ops.geglu_pack_bias(w1.bias.detach().float().contiguous(), f),
                w2=ops.PackedWeight(w2.weight.detach().float().contiguous(), precision=prec), b2=w2.bias.detach().float().contiguous()))
        return packed

    def _needs_autograd(self, x):
        return needs_autograd(self, x)

    def forward(self, x, mask=None):
        """x [b, n, dim]; mask: optional bool [b, n] key-padding mask (True = attend).  Under autograd (training: the reference
        trains its encoders jointly with the denoiser, NS2:1538-1543) the differentiable composite of autograd_path.py runs
        instead of the forward-only HIP kernels."""
        if self.causal:
            raise NotImplementedError("causal=True is not used by any reference caller of Transformer (NS2:252, 315)")
        if self._needs_autograd(x):
            from .autograd_path import transformer_forward_autograd
            return transformer_forward_autograd(self, x, mask)
        return self._forward_hip(x, mask)

    @torch.no_grad()
    def _forward_hip(self, x, mask=None):
        if self.causal:
            raise NotImplementedError("causal=True is not used by any reference caller of Transformer (NS2:252, 315)")
        if self.training and self.dropout > 0:
            raise NotImplementedError("attention dropout is a training-time feature; the HIP path is inference-only")
        b, n, d = x.shape
        prec = _PRECISIONS[self.precision]
        a = self.heads * self.dim_head
        h = x.reshape(b * n, d).float().contiguous()
        for pk in self._pack():
            xn = ops.rmsnorm(h, gamma=pk["g1"], precision=prec)
            qk, vt = ops.linear_qkv(pk["qkv"], xn, seq_len=n, split_col=2 * a, precision=prec)
            o = ops.attention(qk, qk, vt, b, self.heads, n, n, q_col0=0, k_col0=a, precision=prec, key_mask=mask, head_dim=self.dim_head)
            h = ops.linear_f32(pk["out"], o, resid=h, precision=prec)
            xn = ops.rmsnorm(h, gamma=pk["g2"], precision=prec)
            ffh = ops.linear_geglu(pk["w1"], xn, pk["b1"], precision=prec)
            h = ops.linear_f32(pk["w2"], ffh, bias=pk["b2"], resid=h, precision=prec)
        if isinstance(self.norm, _RMSNorm):
            _, h = ops.rmsnorm(h, gamma=self.norm.gamma.detach().float().contiguous(), want_f32=True)
        return h.reshape(b, n, d).to(x.dtype)
