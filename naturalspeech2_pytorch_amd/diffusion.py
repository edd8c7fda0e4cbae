"""`NaturalSpeech2` — the diffusion wrapper around `Model` for the denoising hot path (NS2:1160-1684), with the
reference's own call signatures:

    NaturalSpeech2(model, codec=None, *, timesteps=1000, use_ddim=True, noise_schedule='sigmoid', objective='v', ...)   NS2:1163-1197
    .sample(*, length, prompt=None, batch_size=1, cond_scale=1., text=None, text_lens=None)                              NS2:1457-1466
    .forward(audio, text=None, text_lens=None, mel=None, mel_lens=None, codes=None, prompt=None, pitch=None, ...)       NS2:1503-1515
    .ddim_sample(shape, prompt=None, time_difference=None, cond_scale=1., cond=None)                                     NS2:1380

What runs where.  The per-step path (Model + DDIM update), the codec's RVQ encode/decode and the two conditioning encoders
built from the plain Transformer (`prompt_enc` = SpeechPromptEncoder, `phoneme_enc` = PhonemeEncoder) are HIP.  The rest of
the conditioning front-end — DurationPitchPredictor, Aligner, mel / pitch extraction, the tokenizer (OUT OF SCOPE)
— is not rebuilt: a caller hands over what those modules would have produced through two extra keyword arguments that the
reference signature tolerates (`forward` takes **kwargs):

    cond        [b, dim_prompt, n_c]   the frame-aligned phoneme+pitch conditioning (reference: `expand_encodings`, NS2:1449-1455)
    prompt_enc  [b, n_p, dim_prompt]   the encoded prompt (reference: `self.prompt_enc(prompt)`, NS2:1475 / 1543)

`text`, `text_lens`, `mel`, `mel_lens`, `pitch` are accepted as in the reference; only the branch that would need an
out-of-scope module (no `cond` given) raises, with a NotImplementedError naming it.  `phoneme_enc`, `prompt_enc` and
`pitch_emb` are constructed like upstream (same state_dict keys; load reference checkpoints of a conditional wrapper with
`strict=False`, the out-of-scope members have no counterpart here).

MI355X-first: the per-step elementwise chain of the reference (NS2:1396-1430, ~10 passes over the latents) is ONE fused HIP
kernel (`ns2_ddim_step`), the step-invariant conditioning is computed once per `sample()`, and one denoising step (model +
update) can be captured into a HIP graph and replayed (`use_graph=True`).
"""
import math
from functools import partial

import torch
from torch import nn
import torch.nn.functional as F

from . import ops


def sigmoid_schedule(t, start=-3, end=3, tau=1, clamp_min=1e-9):      # NS2:1144-1148
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    gamma = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    return gamma.clamp(min=clamp_min, max=1.)


def cosine_schedule(t, start=0, end=1, tau=1, clip_min=1e-9):
    """NS2:1136-1142.  Upstream applies `math.cos` to the tensor `t`, which raises for a batch of times (and returns a float
    without `.clamp` for a single one): the schedule cannot run there.  Evaluated with torch.cos, the evident intent."""
    power = 2 * tau
    v_start = math.cos(start * math.pi / 2) ** power
    v_end = math.cos(end * math.pi / 2) ** power
    output = torch.cos((t * (end - start) + start) * math.pi / 2) ** power
    output = (v_end - output) / (v_end - v_start)
    return output.clamp(min=clip_min)


def simple_linear_schedule(t, clip_min=1e-9):                           # NS2:1133-1134
    return (1 - t).clamp(min=clip_min)


_SCHEDULES = {"sigmoid": sigmoid_schedule, "cosine": cosine_schedule, "linear": simple_linear_schedule}


def _safe_div(numer, denom, eps=1e-10):                                  # NS2:1122-1123
    return numer / denom.clamp(min=eps)


def _is_denoiser(m):
    return all(hasattr(m, a) for a in ("forward_with_cond_scale", "dim", "condition_on_prompt")) and isinstance(m, nn.Module)


class NaturalSpeech2(nn.Module):
    def __init__(self, model, codec=None, *, tokenizer=None, target_sample_hz=None, timesteps=1000, use_ddim=True,
                 noise_schedule="sigmoid", objective="v", schedule_kwargs: dict = dict(), time_difference=0.,
                 min_snr_loss_weight=True, min_snr_gamma=5, train_prob_self_cond=0.9, rvq_cross_entropy_loss_weight=0.,
                 dim_codebook: int = 128, duration_pitch_dim: int = 512, aligner_dim_in: int = 80, aligner_dim_hidden: int = 512,
                 aligner_attn_channels: int = 80, num_phoneme_tokens: int = 150, pitch_emb_dim: int = 256,
                 pitch_emb_pp_hidden_dim: int = 512, calc_pitch_with_pyworld=True, mel_hop_length=160,
                 audio_to_mel_kwargs: dict = dict(), scale=1., duration_loss_weight=1., pitch_loss_weight=1.,
                 aligner_loss_weight=1., aligner_bin_loss_weight=0.,
                 encoder_precision="exact"):           # not in the reference: precision mode of the two HIP encoders
        super().__init__()
        assert _is_denoiser(model), "model must be a Model (this package's, or compat.HipBackedModel over the reference's class)"
        self.conditional = model.condition_on_prompt
        self.model = model
        self.codec = codec
        assert codec is not None or target_sample_hz is not None                      # NS2:1207
        self.target_sample_hz = codec.target_sample_hz if codec is not None else target_sample_hz
        self.seq_len_multiple_of = codec.seq_len_multiple_of if codec is not None else None

        if self.conditional:                                                          # NS2:1218-1240, in-scope members only
            from .encoders import PhonemeEncoder, SpeechPromptEncoder
            self.mel_hop_length = mel_hop_length
            self.calc_pitch_with_pyworld = calc_pitch_with_pyworld
            self.phoneme_enc = PhonemeEncoder(tokenizer=tokenizer, num_tokens=num_phoneme_tokens, precision=encoder_precision)
            self.prompt_enc = SpeechPromptEncoder(dim_codebook=dim_codebook, precision=encoder_precision)
            self.pitch_emb = nn.Embedding(pitch_emb_dim, pitch_emb_pp_hidden_dim)
            self.aligner_bin_loss_weight = aligner_bin_loss_weight
            # DurationPitchPredictor, Aligner, AudioToMel, ForwardSumLoss / BinLoss (NS2:1224-1240): out of scope, not built

        assert codec is None or model.dim == codec.codebook_dim, \
            f"transformer model dimension {model.dim} must be equal to codec dimension {codec.codebook_dim}"   # NS2:1244
        self.dim = codec.codebook_dim if codec is not None else model.dim
        assert objective in {"x0", "eps", "v"}, "objective must be either predict x0 or noise"
        if noise_schedule not in _SCHEDULES:
            raise ValueError(f"invalid noise schedule {noise_schedule}")
        assert scale <= 1, "scale must be less than or equal to 1"
        assert use_ddim, "ddpm_sample is unrunnable in the reference (NS2:1361 NameError, NS2:1370 shape); only DDIM is provided"
        self.objective, self.noise_schedule, self.scale = objective, noise_schedule, scale
        self.schedule_kwargs = dict(schedule_kwargs)
        self.gamma_schedule = partial(_SCHEDULES[noise_schedule], **schedule_kwargs)
        self.timesteps, self.use_ddim, self.time_difference = timesteps, use_ddim, time_difference
        self.train_prob_self_cond = train_prob_self_cond
        self.min_snr_loss_weight, self.min_snr_gamma = min_snr_loss_weight, min_snr_gamma
        self.rvq_cross_entropy_loss_weight = rvq_cross_entropy_loss_weight
        self.duration_loss_weight, self.pitch_loss_weight, self.aligner_loss_weight = \
            duration_loss_weight, pitch_loss_weight, aligner_loss_weight

    @property
    def device(self):
        return next(self.model.parameters()).device

    def get_sampling_timesteps(self, batch, *, device):                               # NS2:1303-1308
        times = torch.linspace(1., 0., self.timesteps + 1, device=device)
        times = times[None].expand(batch, -1)
        return [(times[:, i].contiguous(), times[:, i + 1].contiguous()) for i in range(self.timesteps)]

    def _fused_ddim_ok(self):
        """the fused HIP update evaluates the schedule on the device with the reference's default schedule parameters"""
        return not self.schedule_kwargs

    # ------------------------------------------------------------------ sampling (NS2:1379-1431)
    @torch.no_grad()
    def ddim_sample(self, shape, prompt=None, time_difference=None, cond_scale=1., cond=None, noise=None, use_graph=False,
                    on_saturation="demote"):
        """`prompt` here is what the reference passes at NS2:1486-1491: the ENCODED prompt [b, n_p, dim_prompt].

        on_saturation (not in the reference): what to do when a checkpoint's activations leave the IEEE-half range in one of the
        fast precisions (half / mixed / hybrid: conversions clamp at 65504 -- finite but wrong; the model counts them on the
        device).  "demote" (default): the run is REPEATED from the same initial latents with `model.precision = "exact"` (bf16
        planes keep the fp32 exponent range) and the model stays there -- a fast mode never returns clamped audio and never
        needs a hand-picked precision per checkpoint; "raise": Ns2Error, as rounds 2-3 did."""
        assert on_saturation in ("demote", "raise")
        from ._lib import Ns2Error
        batch, device = shape[0], self.device
        start = torch.randn(shape, device=device) if noise is None else noise.to(device).float().clone()
        for attempt in (0, 1):
            if hasattr(self.model, "refresh_weights"):
                self.model.refresh_weights()        # parameters rewritten through `.data` since the last pack (EMA) -> re-pack
            if hasattr(self.model, "clear_cond_cache"):
                self.model.clear_cond_cache()
            try:
                audio = self._ddim_loop(start.clone(), prompt, cond, cond_scale, use_graph)
                if hasattr(self.model, "check_saturation") and audio.is_cuda:
                    # the model polls the device counters every few forwards on its own (any caller); the end of a run takes one
                    # synchronous look (Ns2Error on a new count)
                    self.model.check_saturation(sync=True)
                return audio
            except Ns2Error as e:
                guarded = getattr(self.model, "precision", "exact") in ("half", "mixed", "hybrid", "hybrid_ff")
                if on_saturation == "raise" or attempt == 1 or not guarded or "IEEE-half range" not in str(e):
                    raise
                import warnings
                warnings.warn(f"precision='{self.model.precision}': activations of this checkpoint leave the IEEE-half range; repeating "
                              f"the sampling run with precision='exact' and keeping the model there ({e})")
                self.model.precision = "exact"

    def _ddim_loop(self, audio, prompt, cond, cond_scale, use_graph):
        batch, device = audio.shape[0], audio.device
        pairs = self.get_sampling_timesteps(batch, device=device)
        # `time_difference` only shifts times_next AFTER gamma_next was taken (NS2:1396-1406): it has no effect upstream either
        if not self._fused_ddim_ok():
            return self._ddim_sample_unfused(audio, pairs, prompt, cond, cond_scale)
        # the run's times are known here and shared by the batch, so every time-conditioning projection of the run is
        # computed once, ahead of the loop; step i reads row i (Model.time_table)
        table = None
        if hasattr(self.model, "time_table") and audio.is_cuda:
            table = self.model.time_table(torch.stack([p[0][0] for p in pairs]), batch)
        if use_graph:
            return self._ddim_sample_graph(audio, pairs, prompt, cond, cond_scale, table)
        for i, (times, times_next) in enumerate(pairs):
            kw = {} if table is None else dict(cond_row=table[i])
            out = self.model.forward_with_cond_scale(audio, times, prompt=prompt, cond_scale=cond_scale, cond=cond, **kw)
            ops.ddim_step(audio, out, times, times_next, self.objective, self.noise_schedule, self.scale, out=audio)
        return audio

    def _ddim_sample_unfused(self, audio, pairs, prompt, cond, cond_scale):
        """non-default `schedule_kwargs`: the schedule is evaluated by the host-side functions above, the model step is HIP"""
        for times, times_next in pairs:
            g, gn = self.gamma_schedule(times)[:, None, None], self.gamma_schedule(times_next)[:, None, None]
            alpha, sigma = torch.sqrt(g) * self.scale, torch.sqrt(1 - g)
            alpha_n, sigma_n = torch.sqrt(gn) * self.scale, torch.sqrt(1 - gn)
            out = self.model.forward_with_cond_scale(audio, times, prompt=prompt, cond_scale=cond_scale, cond=cond)
            x0 = {"x0": lambda: out, "eps": lambda: _safe_div(audio - sigma * out, alpha),
                  "v": lambda: alpha * audio - sigma * out}[self.objective]()
            eps = _safe_div(audio - alpha * x0, sigma)
            audio = x0 * alpha_n + eps * sigma_n
        return audio

    def _ddim_sample_graph(self, audio, pairs, prompt, cond, cond_scale, table=None):
        """one (model + DDIM update) step captured in a HIP graph; times (and the step's row of the conditioning table) are device
        buffers rewritten per step."""
        t_buf, tn_buf = pairs[0][0].clone(), pairs[0][1].clone()
        row = table[0].clone() if table is not None else None
        kw = {} if row is None else dict(cond_row=row)
        step = lambda: ops.ddim_step(                                                   # noqa: E731
            audio, self.model.forward_with_cond_scale(audio, t_buf, prompt=prompt, cond_scale=cond_scale, cond=cond, **kw),
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
        for i, (times, times_next) in enumerate(pairs):
            t_buf.copy_(times)
            tn_buf.copy_(times_next)
            if row is not None:
                row.copy_(table[i])
            g.replay()
        return audio

    # ------------------------------------------------------------------ conditioning (NS2:1433-1455, 1474-1483)
    def process_prompt(self, prompt=None):                                            # NS2:1433-1447
        if prompt is None:
            return None
        assert self.model.condition_on_prompt
        is_raw_prompt = prompt.ndim == 2
        assert not (is_raw_prompt and self.codec is None), "codec must be passed in if one were to train on raw prompt"
        if is_raw_prompt:
            with torch.no_grad():
                self.codec.eval()
                prompt, _, _ = self.codec(prompt, curtail_from_left=True, return_encoded=True)
        return prompt

    def _encode_prompt(self, prompt, prompt_enc):
        if prompt_enc is not None:
            return prompt_enc
        assert prompt is not None, "conditional model: pass `prompt` (raw audio or codec latents) or `prompt_enc`"
        return self.prompt_enc(self.process_prompt(prompt))                            # NS2:1474-1475 / 1542-1543 (HIP encoder)

    def refresh_weights(self):
        """run boundary: every HIP module re-checks its packed weights against the parameters' CONTENT (writes through `.data` --
        what ema_pytorch does to the copy of this whole object it samples from, NS2:1793-1801 -- bump no version counter)"""
        for name in ("model", "prompt_enc", "phoneme_enc", "codec"):
            mod = getattr(self, name, None)
            if mod is not None and hasattr(mod, "refresh_weights"):
                mod.refresh_weights()

    @torch.no_grad()
    def sample(self, *, length, prompt=None, batch_size=1, cond_scale=1., text=None, text_lens=None,
               cond=None, prompt_enc=None, noise=None, use_graph=False):
        """NS2:1457-1501.  Extra keywords (not in the reference): `cond` / `prompt_enc` = pre-computed conditioning (module
        docstring), `noise` = injected initial latents (parity tests), `use_graph` = HIP-graph replay of the step."""
        self.refresh_weights()
        p_enc = None
        if self.conditional:
            assert (prompt is not None or prompt_enc is not None) and (text is not None or cond is not None)   # NS2:1473
            p_enc = self._encode_prompt(prompt, prompt_enc)
            if cond is None:
                raise NotImplementedError(
                    "sample(text=...) derives the aligned conditioning with the DurationPitchPredictor (NS2:1478-1483), which is "
                    "outside the HIP hot path: pass the aligned conditioning as `cond=` [b, dim_prompt, n_frames]")
            batch_size = p_enc.shape[0]
        elif prompt is not None:
            batch_size = prompt.shape[0]                                               # NS2:1485-1486
        audio = self.ddim_sample((batch_size, length, self.dim), prompt=p_enc, cond=cond, cond_scale=cond_scale, noise=noise,
                                 use_graph=use_graph)
        if self.codec is not None:
            audio = self.codec.decode(audio)
            if audio.ndim == 3:
                audio = audio[:, 0]
        return audio

    # ------------------------------------------------------------------ training loss (NS2:1503-1684)
    def forward(self, audio, text=None, text_lens=None, mel=None, mel_lens=None, codes=None, prompt=None, pitch=None,
                *args, cond=None, prompt_enc=None, times=None, noise=None, **kwargs):
        """Reference signature; `cond`, `prompt_enc` (module docstring) and `times`, `noise` (deterministic parity tests) are
        the keyword-only extras.  The model call runs the autograd composite when gradients are required."""
        is_raw = audio.ndim == 2
        p_enc = None
        if self.conditional:
            p_enc = self._encode_prompt(prompt, prompt_enc)
            if cond is None:
                raise NotImplementedError(
                    "forward(text=..., mel=..., pitch=...) derives the aligned conditioning with the Aligner, the "
                    "DurationPitchPredictor and mel / pitch extraction (NS2:1524-1602), which are outside the HIP hot path: pass "
                    "the aligned conditioning as `cond=` [b, dim_prompt, n_frames]")
        assert not (is_raw and self.codec is None), "codec must be passed in if one were to train on raw audio"
        if is_raw:
            with torch.no_grad():
                self.codec.eval()
                audio, codes, _ = self.codec(audio, return_encoded=True)             # NS2:1608-1611 (RVQ encode in HIP)
        batch, n, d = audio.shape
        assert d == self.dim, f"codec codebook dimension {d} must match model dimensions {self.dim}"
        device = audio.device
        times = torch.zeros((batch,), device=device).float().uniform_(0, 1.) if times is None else times
        noise = torch.randn_like(audio) if noise is None else noise
        gamma = self.gamma_schedule(times)[:, None, None]
        alpha, sigma = torch.sqrt(gamma) * self.scale, torch.sqrt(1 - gamma)         # NS2:1152-1153
        noised = alpha * audio + sigma * noise
        pred = self.model(noised, times, prompt=p_enc, cond=cond)                     # NS2:1635
        target = {"eps": noise, "x0": audio, "v": alpha * noise - sigma * audio}[self.objective]
        loss = F.mse_loss(pred, target, reduction="none").flatten(1).mean(dim=1)      # [b]
        snr = (alpha * alpha) / (sigma * sigma)                                       # [b, 1, 1]
        clipped = snr.clamp(max=self.min_snr_gamma) if self.min_snr_loss_weight else snr
        weight = {"eps": clipped / snr, "x0": clipped, "v": clipped / (snr + 1)}[self.objective]
        # NS2:1668 as written upstream: `loss` is [b] and `loss_weight` is [b, 1, 1], so the product broadcasts to
        # [b, 1, b] and the mean equals mean(loss) * mean(weight) -- kept bit-compatible with the reference.
        loss = (loss * weight).mean()
        if self.rvq_cross_entropy_loss_weight == 0 or codes is None:                  # NS2:1672-1673
            return loss
        x_start = {"x0": lambda: pred, "eps": lambda: _safe_div(audio - sigma * pred, alpha),
                   "v": lambda: alpha * audio - sigma * pred}[self.objective]()
        assert self.codec is not None and hasattr(self.codec, "rq"), "the RVQ cross-entropy term needs codec.rq (NS2:1682)"
        _, ce_loss = self.codec.rq(x_start, codes)
        return loss + self.rvq_cross_entropy_loss_weight * ce_loss                    # NS2:1684 (duration_pitch_loss is 0 upstream)
