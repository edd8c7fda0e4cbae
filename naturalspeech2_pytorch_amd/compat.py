"""Drop-in at the reference's OWN boundary: a subclass of the reference's `Model` class whose inference forward runs in
libns2hip, so that the unmodified reference `NaturalSpeech2` / `Trainer` can hold it:

    import naturalspeech2_pytorch as ref                       # the upstream package
    from naturalspeech2_pytorch_amd.compat import hip_backed_model_class
    HipBackedModel = hip_backed_model_class(ref.Model)         # subclass of ref.Model: passes `model: Model` (beartype, NS2:1165)
    model = HipBackedModel(dim=128, depth=6).cuda()            # same constructor keywords (NS2:814-831) + precision=
    diffusion = ref.NaturalSpeech2(model=model, codec=codec, timesteps=1000)
    audio = diffusion.sample(length=1024)                      # NS2:1410 -> model.forward_with_cond_scale -> HIP
    loss = diffusion(raw_audio); loss.backward()               # NS2:1635 -> ref.Model.forward (autograd stays upstream's)

The subclass owns the reference's parameters (constructed by the reference's own `__init__`, hence identical
initialisation and state_dict), and `HipDenoiserMixin` hands them to the library by the same key names
(SURVEY §8b state_dict contract).  Calls that need autograd, or per-utterance stochastic conditioning dropout, go to the
reference's own `forward` — the composite of `autograd_path.py` is not involved here.

The codec side of the same boundary (round 6): the reference type-checks `codec: Optional[Union[SoundStream, EncodecWrapper]]`
(NS2:1166) and then only uses the surface of SURVEY §8b (NS2:1212-1214, 1244, 1445, 1496, 1611, 1682):

    from audiolm_pytorch import EncodecWrapper                 # the class the reference imports (NS2:23)
    from naturalspeech2_pytorch_amd.compat import hip_backed_codec_class
    HipEncodec = hip_backed_codec_class(EncodecWrapper)        # subclass of EncodecWrapper: passes the type check
    codec = HipEncodec.from_hf(transformers.EncodecModel(...)) # or HipEncodec(codebooks, encoder=, decoder=)
    diffusion = ref.NaturalSpeech2(model=model, codec=codec)   # RVQ encode / decode + SEANet run in libns2hip
"""
import inspect

from .model import HipDenoiserMixin, _CFG_KEYS


def hip_backed_model_class(reference_model_cls):
    """returns `HipBackedModel(reference_model_cls)`; the reference class is passed in so that this package never imports
    (or depends on) the upstream package itself"""
    sig = inspect.signature(reference_model_cls.__init__)
    defaults = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}

    class HipBackedModel(HipDenoiserMixin, reference_model_cls):
        def __init__(self, dim, *args, precision="exact", train_backend="reference", train_precision="exact", **kwargs):
            reference_model_cls.__init__(self, dim, *args, **kwargs)
            bound = sig.bind(self, dim, *args, **kwargs)
            cfg = dict(defaults)
            cfg.update({k: v for k, v in bound.arguments.items() if k != "self"})
            self._hip_init({k: cfg[k] for k in _CFG_KEYS}, precision)
            assert train_backend in ("reference", "hip") and train_precision in ("exact", "mixed")
            self.train_backend = train_backend
            self.train_precision = train_precision            # arithmetic of train_backend="hip" (training.py)

        def _forward_autograd(self, x, times, prompt=None, prompt_mask=None, cond=None, cond_drop_prob=None):
            """train_backend="reference" (default): the reference's own forward under torch autograd -- the loss is upstream's bit for
            bit.  "hip": forward AND backward in libns2hip (training.py), through the reference's own parameters; the unmodified
            reference `Trainer` / `NaturalSpeech2.forward` (NS2:1635, 1886) then trains on the HIP kernels."""
            if self.train_backend == "hip" and x.is_cuda and prompt_mask is None:
                # (a prompt_mask goes to the reference's own forward below, which raises on it: model.py _PROMPT_MASK_MSG)
                from . import training
                why = training.unsupported_reason(self)
                if why is None:
                    return training.model_forward_train(self, x, times, prompt=prompt, cond=cond, cond_drop_prob=cond_drop_prob)
                import warnings
                warnings.warn(f"HIP training path not usable for this Model ({why}): running the reference's own forward")
            return reference_model_cls.forward(self, x, times, prompt=prompt, prompt_mask=prompt_mask, cond=cond,
                                               cond_drop_prob=cond_drop_prob)

    HipBackedModel.__qualname__ = HipBackedModel.__name__ = "HipBackedModel"
    return HipBackedModel


def hip_backed_codec_class(reference_codec_cls):
    """returns `HipBackedEncodec`, a subclass of BOTH `reference_codec_cls` (audiolm_pytorch's `EncodecWrapper`, or `SoundStream`:
    whatever the reference's `codec:` annotation accepts, NS2:1166) and this package's `EncodecWrapperHIP`, whose residual-VQ
    encode / decode (csrc/rvq.hip) and SEANet encoder / decoder (seanet.py) run on the HIP kernels.  The reference class's own
    `__init__` is NOT run (audiolm's downloads the pretrained EnCodec checkpoint and builds the PyTorch codec this class replaces):
    construct it like `EncodecWrapperHIP` -- `HipBackedEncodec(codebooks, encoder=, decoder=)` or `.from_hf(hf_encodec_model)`.
    Everything the reference touches comes from `EncodecWrapperHIP`: target_sample_hz / seq_len_multiple_of / codebook_dim, forward(x,
    return_encoded=, curtail_from_left=) -> (emb, codes, None), decode(emb), rq(x, codes) (codec.py)."""
    from torch import nn

    from .codec import EncodecWrapperHIP

    class HipBackedEncodec(EncodecWrapperHIP, reference_codec_cls):
        def __init__(self, codebooks, encoder=None, decoder=None):
            nn.Module.__init__(self)
            self._hip_codec_init(codebooks, encoder, decoder)

    HipBackedEncodec.__qualname__ = HipBackedEncodec.__name__ = "HipBackedEncodec"
    return HipBackedEncodec
