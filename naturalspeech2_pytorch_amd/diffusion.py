"""`NaturalSpeech2` — the diffusion wrapper around `Model` for the denoising hot path (NS2:1160-1684).

Kept from the reference: constructor keywords that shape the sampler (timesteps, use_ddim, noise_schedule,
objective, scale, time_difference, min_snr_*), `.ddim_sample(shape, prompt=, cond_scale=, cond=)` NS2:1380,
`.sample(length=, prompt=, batch_size=, cond_scale=, ...)` NS2:1458 and `.forward(audio, ...) -> loss` NS2:1503.
Out of scope: the conditioning front-end (PhonemeEncoder, SpeechPromptEncoder, DurationPitchPredictor,
Aligner, mel/pitch extraction).  A conditional model is therefore driven with the already-encoded `prompt`
[b, n_p, dim_prompt] and aligned `cond` [b, dim_prompt, n_c] — the two tensors the reference hands to
`Model.forward` at NS2:1410 / NS2:1635.

MI355X-first: the per-step elementwise chain of the reference (NS2:1396-1430, ~10 passes over the latents) is ONE
fused HIP kernel (`ns2_ddim_step`), the step-invariant conditioning is computed once per `sample()`, and one
denoising step (model + update) can be captured into a HIP graph and replayed (`use_graph=True`).
"""
import math
from typing import Optional

import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from .model import Model


def sigmoid_schedule(t, start=-3, end=3, tau=1, clamp_min=1e-9):      # NS2:1144-1148
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    gamma = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    return gamma.clamp(min=clamp_min, max=1.)


def cosine_schedule(t, clip_min=1e-9):                                  # NS2:1136-1142 with defaults
    return (torch.cos(t * math.pi / 2) ** 2).clamp(min=clip_min)


def simple_linear_schedule(t, clip_min=1e-9):                           # NS2:1133-1134
    return (1 - t).clamp(min=clip_min)


_SCHEDULES = {"sigmoid": sigmoid_schedule, "cosine": cosine_schedule, "linear": simple_linear_schedule}


class NaturalSpeech2(nn.Module):
    def __init__(self, model: Model, codec=None, *, target_sample_hz=None, timesteps=1000, use_ddim=True,
                 noise_schedule="sigmoid", objective="v", time_difference=0., min_snr_loss_weight=True, min_snr_gamma=5,
                 rvq_cross_entropy_loss_weight=0., scale=1., **unused_frontend_kwargs):
        super().__init__()
        assert isinstance(model, Model)
        self.conditional = model.condition_on_prompt
        self.model = model
        self.codec = codec
        assert codec is not None or target_sample_hz is not None                      # NS2:1207
        self.target_sample_hz = codec.target_sample_hz if codec is not None else target_sample_hz
        self.seq_len_multiple_of = codec.seq_len_multiple_of if codec is not None else None
        assert codec is None or model.dim == codec.codebook_dim, \
            f"transformer model dimension {model.dim} must be equal to codec dimension {codec.codebook_dim}"   # NS2:1244
        self.dim = codec.codebook_dim if codec is not None else model.dim
        assert objective in {"x0", "eps", "v"}
        assert noise_schedule in _SCHEDULES, f"invalid noise schedule {noise_schedule}"
        assert scale <= 1
        assert use_ddim, "ddpm_sample is unrunnable in the reference (NS2:1361 NameError); only DDIM is provided"
        self.objective, self.noise_schedule, self.scale = objective, noise_schedule, scale
        self.gamma_schedule = _SCHEDULES[noise_schedule]
        self.timesteps, self.use_ddim, self.time_difference = timesteps, use_ddim, time_difference
        self.min_snr_loss_weight, self.min_snr_gamma = min_snr_loss_weight, min_snr_gamma
        self.rvq_cross_entropy_loss_weight = rvq_cross_entropy_loss_weight
        self._graph = None

    @property
    def device(self):
        return next(self.model.parameters()).device

    def get_sampling_timesteps(self, batch, *, device):                               # NS2:1303-1308
        times = torch.linspace(1., 0., self.timesteps + 1, device=device)
        times = times[None].expand(batch, -1)
        return [(times[:, i].contiguous(), times[:, i + 1].contiguous()) for i in range(self.timesteps)]

    # ------------------------------------------------------------------ sampling (NS2:1379-1431)
    @torch.no_grad()
    def ddim_sample(self, shape, prompt=None, time_difference=None, cond_scale=1., cond=None, noise=None, use_graph=False):
        batch, device = shape[0], self.device
        audio = torch.randn(shape, device=device) if noise is None else noise.to(device).float().clone()
        pairs = self.get_sampling_timesteps(batch, device=device)
        if use_graph:
            return self._ddim_sample_graph(audio, pairs, prompt, cond, cond_scale)
        for times, times_next in pairs:
            out = self.model.forward_with_cond_scale(audio, times, prompt=prompt, cond_scale=cond_scale, cond=cond)
            ops.ddim_step(audio, out, times, times_next, self.objective, self.noise_schedule, self.scale, out=audio)
        return audio

    def _ddim_sample_graph(self, audio, pairs, prompt, cond, cond_scale):
        """one (model + DDIM update) step captured in a HIP graph; times are device tensors rewritten per step."""
        t_buf, tn_buf = pairs[0][0].clone(), pairs[0][1].clone()
        step = lambda: ops.ddim_step(                                                   # noqa: E731
            audio, self.model.forward_with_cond_scale(audio, t_buf, prompt=prompt, cond_scale=cond_scale, cond=cond),
            t_buf, tn_buf, self.objective, self.noise_schedule, self.scale, out=audio)
        keep = audio.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm-up outside capture: packs weights, sizes workspaces
            step()
        torch.cuda.current_stream().wait_stream(side)
        audio.copy_(keep)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        audio.copy_(keep)
        for times, times_next in pairs:
            t_buf.copy_(times)
            tn_buf.copy_(times_next)
            g.replay()
        return audio

    @torch.no_grad()
    def sample(self, *, length, prompt=None, batch_size=1, cond_scale=1., text=None, text_lens=None, cond=None, noise=None,
               use_graph=False):
        """NS2:1457-1501.  Conditional models take the encoded `prompt` and aligned `cond` (see module docstring)."""
        if self.conditional:
            if text is not None:
                raise NotImplementedError("text front-end (PhonemeEncoder/DurationPitchPredictor, NS2:1474-1483) is out of "
                                          "scope of the HIP hot path; pass the encoded `prompt` and aligned `cond` tensors")
            assert prompt is not None and cond is not None
            batch_size = prompt.shape[0]
        audio = self.ddim_sample((batch_size, length, self.dim), prompt=prompt, cond=cond, cond_scale=cond_scale, noise=noise,
                                 use_graph=use_graph)
        if self.codec is not None:
            audio = self.codec.decode(audio)
            if audio.ndim == 3:
                audio = audio[:, 0]
        return audio

    # ------------------------------------------------------------------ training loss (NS2:1503-1684)
    def forward(self, audio, prompt=None, cond=None, codes=None, times=None, noise=None):
        is_raw = audio.ndim == 2
        assert not (is_raw and self.codec is None), "codec must be passed in if one were to train on raw audio"
        if is_raw:
            with torch.no_grad():
                self.codec.eval()
                audio, codes, _ = self.codec(audio, return_encoded=True)             # NS2:1608-1611 (RVQ encode in HIP)
        batch, n, d = audio.shape
        assert d == self.dim
        device = audio.device
        times = torch.zeros((batch,), device=device).float().uniform_(0, 1.) if times is None else times
        noise = torch.randn_like(audio) if noise is None else noise
        gamma = self.gamma_schedule(times)[:, None, None]
        alpha, sigma = torch.sqrt(gamma) * self.scale, torch.sqrt(1 - gamma)         # NS2:1152-1153
        noised = alpha * audio + sigma * noise
        pred = self.model(noised, times, prompt=prompt, cond=cond)                    # NS2:1635
        target = {"eps": noise, "x0": audio, "v": alpha * noise - sigma * audio}[self.objective]
        loss = F.mse_loss(pred, target, reduction="none").flatten(1).mean(dim=1)
        snr = ((alpha * alpha) / (sigma * sigma)).flatten()
        clipped = snr.clamp(max=self.min_snr_gamma) if self.min_snr_loss_weight else snr
        weight = {"eps": clipped / snr, "x0": clipped, "v": clipped / (snr + 1)}[self.objective]
        return (loss * weight).mean()                                                 # NS2:1668
