"""Differentiable PyTorch composite of `Model.forward`: fp32 torch ops on the module's own parameters.

Since round 4 training has HIP kernels of its own (`training.py`: forward AND backward in libns2hip, `train_backend="hip"`, the
default of this package's `Model` on an MI355X; NS2:1635, NS2:1886).  This composite is what is left for the cases those kernels do
not take -- CPU tensors (the CPU test-suite), `train_backend="composite"`, a model `training.unsupported_reason` rejects (a head
dimension other than 64 in the backward kernels, non-fp32 parameters) -- and the yardstick bench.py times beside the HIP training
step (`side.train_step.*.pytorch_composite_ms_per_step`).  Inference (`torch.no_grad()` -- `NaturalSpeech2.sample`,
`forward_with_cond_scale` in the sampling loop, bench.py's timed region) never reaches this file.
"""
import math

import torch
import torch.nn.functional as F


def _causal_conv(x_bnc, conv, dilation=1):
    k = conv.weight.shape[-1]
    x = F.pad(x_bnc.transpose(1, 2), (dilation * (k - 1), 0))
    return F.conv1d(x, conv.weight, conv.bias, dilation=dilation).transpose(1, 2)


def _rmsnorm(x, norm, t=None):
    out = F.normalize(x, dim=-1) * math.sqrt(x.shape[-1])
    if norm.gamma is not None:
        out = out * norm.gamma
    if norm.to_gamma_beta is None:
        return out
    g, b = norm.to_gamma_beta(t).chunk(2, dim=-1)
    return out * g[:, None] + b[:, None]


def _attention(x, attn, heads, context=None, include_queries=False, key_mask=None, dropout_p=0.):
    ctx = x if context is None else (torch.cat((x, context), dim=1) if include_queries else context)
    q = attn.to_q(x)
    k, v = attn.to_kv(ctx).chunk(2, dim=-1)
    b, n, _ = q.shape

    def sp(t):
        return t.reshape(b, t.shape[1], heads, -1).transpose(1, 2)

    am = None if key_mask is None else key_mask[:, None, None, :].bool()          # ATT:92-94: key-padding mask, True = attend
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=am, dropout_p=dropout_p)
    return attn.to_out(o.transpose(1, 2).reshape(b, n, -1))


def _feedforward(x, ff, causal_conv):
    h = getattr(ff, "0")(x)
    a, gate = h.chunk(2, dim=-1)
    h = F.gelu(gate) * a
    if causal_conv:
        h = _causal_conv(h, getattr(getattr(ff, "2"), "1"))
        return getattr(ff, "3")(h)
    return getattr(ff, "2")(h)


def model_forward_autograd(m, x, times, prompt=None, cond=None, cond_drop_prob=None):
    b, n, _ = x.shape
    p = m.cond_drop_prob if cond_drop_prob is None else cond_drop_prob
    w = getattr(m.to_time_cond, "0").weights
    tt = times[:, None]
    fr = tt * w[None] * 2 * math.pi
    t = F.silu(getattr(m.to_time_cond, "1")(torch.cat((tt, fr.sin(), fr.cos()), dim=-1)))
    c = None
    h = x
    if m.condition_on_prompt:
        assert prompt is not None and cond is not None

        def mask():
            if p == 1:
                return torch.ones(b, dtype=torch.bool, device=x.device)
            if p == 0:
                return torch.zeros(b, dtype=torch.bool, device=x.device)
            return torch.rand(b, device=x.device) < p

        dm = mask()
        pc = F.silu(getattr(m.to_prompt_cond, "1")(prompt.mean(dim=1)))
        pc = torch.where(dm[:, None], m.null_prompt_cond, pc)
        t = torch.cat((t, pc), dim=-1)
        pr = m.perceiver_resampler
        px = pr.proj_context(prompt) if hasattr(pr, "proj_context") else prompt
        lat = pr.latents[None].expand(b, -1, -1)
        for attn, ff in pr.layers:
            lat = _attention(lat, attn, m.heads, context=px, include_queries=True) + lat
            lat = _feedforward(lat, ff, False) + lat
        c = torch.where(dm[:, None, None], m.null_prompt_tokens, _rmsnorm(lat, pr.norm))
        cm = F.conv1d(cond, m.cond_to_model_dim.weight, m.cond_to_model_dim.bias)
        cm = torch.where(mask()[:, None, None], m.null_cond, cm)
        if cm.shape[-1] > n:
            cm = cm[..., :n]
        elif cm.shape[-1] < n:
            cm = F.pad(cm, (0, n - cm.shape[-1]))
        h = h + cm.transpose(1, 2)
    wn = m.wavenet
    h0 = _causal_conv(h, wn.init_conv)
    cols = [h0] * m.wavenet_layers
    skips = []
    for s, st in enumerate(wn.stacks):
        nxt = []
        for i, blk in enumerate(st.blocks):
            u = cols[i]
            g, be = blk.to_time_cond(t).chunk(2, dim=-1)
            z = _causal_conv(u, blk.conv, 2 ** i) * g[:, None] + be[:, None]
            z = z.tanh() * z.sigmoid() + _causal_conv(u, blk.res_conv)
            nxt.append(z)
            if blk.skip_conv is not None:
                skips.append(_causal_conv(z, blk.skip_conv))
        cols = nxt
    h = _causal_conv(torch.stack(skips).sum(0), wn.final_conv)
    for layer in m.transformer.layers:
        h = _attention(_rmsnorm(h, getattr(layer, "0"), t), getattr(layer, "1"), m.heads) + h
        if m.condition_on_prompt:
            h = _attention(_rmsnorm(h, getattr(layer, "2"), t), getattr(layer, "3"), m.heads, context=c) + h
        h = _feedforward(_rmsnorm(h, getattr(layer, "4"), t), getattr(layer, "5"), True) + h
    tp = m.transformer.to_pred
    return getattr(tp, "1")(_rmsnorm(h, getattr(tp, "0")))


# ---- the plain Transformer and the two conditioning encoders under autograd (NS2:1073-1115, 228-341): joint training of
# prompt_enc / phoneme_enc with the denoiser (NS2:1538-1543) needs gradients through them; the HIP forwards are inference-only.
def transformer_forward_autograd(tr, x, mask=None):
    p = tr.dropout if tr.training else 0.
    for norm1, attn, norm2, ff in tr.layers:
        x = _attention(_rmsnorm(x, norm1), attn, tr.heads, key_mask=mask, dropout_p=p) + x
        x = _feedforward(_rmsnorm(x, norm2), ff, False) + x
    return _rmsnorm(x, tr.norm) if hasattr(tr.norm, "gamma") else x


def speech_prompt_encoder_autograd(enc, x):
    h = x.transpose(1, 2)
    for m in enc.conv:
        if isinstance(m, torch.nn.Conv1d):
            h = F.silu(F.conv1d(h, m.weight, m.bias, padding=enc.padding))
    return transformer_forward_autograd(enc.transformer, h.transpose(1, 2))


def phoneme_encoder_autograd(enc, ids, mask=None):
    ids = ids.masked_fill(ids < 0, enc.pad_id)
    h = F.embedding(ids, enc.token_emb.weight)
    h = F.silu(_causal_conv(h, enc.conv[1]))
    h = F.dropout(h, enc.conv_dropout, enc.training)
    return transformer_forward_autograd(enc.transformer, h, mask=mask)
