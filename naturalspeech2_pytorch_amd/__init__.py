"""MI355X-native NaturalSpeech2 denoising hot path (drop-in for lucidrains/naturalspeech2-pytorch's
`Model`, `NaturalSpeech2.sample/.forward` and the EnCodec RVQ encode).  See DESIGN.md."""
from ._lib import Ns2Error, load as load_library          # noqa: F401
from .model import Model                                   # noqa: F401
from .diffusion import NaturalSpeech2                      # noqa: F401
from .codec import EncodecWrapperHIP, HipRVQ               # noqa: F401
from .transformer import Transformer                        # noqa: F401
from .encoders import PhonemeEncoder, SpeechPromptEncoder   # noqa: F401
from .seanet import SEANetDecoderHIP, SEANetEncoderHIP      # noqa: F401

__all__ = ["Model", "NaturalSpeech2", "Transformer", "PhonemeEncoder", "SpeechPromptEncoder", "EncodecWrapperHIP", "HipRVQ", "SEANetEncoderHIP", "SEANetDecoderHIP", "Ns2Error", "load_library"]
