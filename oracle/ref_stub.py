"""TEST INFRASTRUCTURE ONLY (oracle/): import the *unmodified* reference from /root/reference.

The reference package (`naturalspeech2_pytorch/__init__.py:8-24`,
`naturalspeech2_pytorch.py:17-40`) imports torchaudio, audiolm_pytorch, beartype, ema_pytorch,
pyworld, inflect, num2words, ... none of which are installed here and none of which are touched
by the arithmetic of `Model` / `NaturalSpeech2.ddim_sample`.  We pre-populate `sys.modules`
with inert stand-ins and then execute the reference's own source files unchanged.

Where /root/reference exists (the build container) the reference is imported from there.  On the
GPU box it is not; there `oracle/_ref/reference_py.tar.gz` (written by `oracle/make_ref.py` from the
untouched reference sources; git-ignored, travels with the snapshot) is unpacked into a temporary
directory and imported from it.  Used by `tests/golden/make_golden.py` (fixture generation), by the
tests that pin `oracle/ns2_oracle.py` against the real reference, by `tests/test_reference_gpu.py`
(the reference's own `NaturalSpeech2` driving the HIP model) and by `bench.py`'s `cpu_baseline`.
Never imported by the product package.
"""
import importlib
import os
import sys
import types
import typing

REFERENCE_ROOT = os.environ.get("NS2_REFERENCE_ROOT", "/root/reference")
ARCHIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference_py.tar.gz")
_unpacked = None


def reference_source() -> str:
    """"live" (the reference checkout), "archive" (oracle/_ref, see oracle/make_ref.py) or "" (absent)"""
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "naturalspeech2_pytorch")):
        return "live"
    return "archive" if os.path.isfile(ARCHIVE) else ""


def reference_available() -> bool:
    return reference_source() != ""


def _reference_root() -> str:
    global _unpacked
    if reference_source() == "live":
        return REFERENCE_ROOT
    if _unpacked is None:
        import atexit
        import shutil
        import tarfile
        import tempfile
        _unpacked = tempfile.mkdtemp(prefix="ns2_ref_")
        atexit.register(shutil.rmtree, _unpacked, ignore_errors=True)
        with tarfile.open(ARCHIVE, "r:gz") as tf:
            tf.extractall(_unpacked)
    return _unpacked


def _mod(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    # a spec, so that `importlib.util.find_spec(name)` (transformers probes torchaudio that way) answers instead of raising
    # "ValueError: torchaudio.__spec__ is None"; there is no distribution metadata, so availability probes still say "absent"
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    sys.modules[name] = m
    return m


def _install_stubs():
    import torch
    from torch import nn

    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    if "torchaudio" not in sys.modules:
        ta = _mod("torchaudio")
        ta.transforms = _mod("torchaudio.transforms", Spectrogram=_Dummy, MelScale=_Dummy, Resample=_Dummy)
        ta.functional = _mod("torchaudio.functional")
    if "audiolm_pytorch" not in sys.modules:
        al = _mod("audiolm_pytorch", SoundStream=type("SoundStream", (_Dummy,), {}),
                  EncodecWrapper=type("EncodecWrapper", (_Dummy,), {}))
        al.data = _mod("audiolm_pytorch.data", SoundDataset=object, get_dataloader=lambda *a, **k: None)
    if "beartype" not in sys.modules:
        bt = _mod("beartype", beartype=lambda f: f)
        bt.typing = _mod("beartype.typing", **{k: getattr(typing, k) for k in
                                               ("Tuple", "Union", "Optional", "List", "Dict", "Callable", "Any")})
        def _is_bearable(obj, hint):            # only use in the reference: is_bearable(x, List[str]) at NS2:277
            return isinstance(obj, (list, tuple)) and all(isinstance(e, str) for e in obj)
        bt.door = _mod("beartype.door", is_bearable=_is_bearable)
    if "ema_pytorch" not in sys.modules:
        _mod("ema_pytorch", EMA=_Dummy)
    for name in ("pyworld", "inflect", "num2words", "num_to_words"):
        if name not in sys.modules:
            m = _mod(name)
            m.engine = lambda *a, **k: None          # inflect.engine()
            m.num2words = lambda *a, **k: ""
            m.num_to_word = lambda *a, **k: ""


_cached = None


def load_reference():
    """Returns the reference module `naturalspeech2_pytorch.naturalspeech2_pytorch` (source unmodified)."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError(f"reference present neither at {REFERENCE_ROOT} nor as {ARCHIVE} (oracle/make_ref.py)")
    _install_stubs()
    root = _reference_root()
    if root not in sys.path:
        sys.path.insert(0, root)
    _cached = importlib.import_module("naturalspeech2_pytorch.naturalspeech2_pytorch")
    return _cached
