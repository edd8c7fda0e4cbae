"""Round-2 parity hardening (VERDICT r1 "next" #1, #3, #4, #5, #6, #8): the benched precision modes under sweeps (weight
seeds x diffusion times at the headline architecture), conditioned + classifier-free guidance, a 50-step DDIM trajectory,
a large-activation stress, the mixed-mode plane format, repack after `.data` writes, the reference's sampling loop as the
caller of the HIP model, the non-default schedules / objectives, concurrent models on separate streams (and devices when
there are two), the sampler over an RCCL process group, RVQ at BASELINE config-4 size with every mismatch adjudicated,
and the codec boundary class with HF EnCodec's SEANet injected.

Measured numbers are merged key by key into the parity record (tests/parity_record.py -> profiles/r06_parity.json)."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from naturalspeech2_pytorch_amd import EncodecWrapperHIP, HipRVQ, Model, NaturalSpeech2, ops  # noqa: E402
from naturalspeech2_pytorch_amd import distributed as D  # noqa: E402
from oracle import ns2_oracle as O  # noqa: E402
from oracle import rvq_oracle as R  # noqa: E402
from tests.golden.gen import make_input, make_weights  # noqa: E402

DEV = torch.device("cuda:0")
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 1e-3                                  # BASELINE.json north_star: <= 1e-3 relative to the fp32 reference
# asserted ceilings per mode: "half" only has to meet the tolerance; "hybrid" (the benched mode: "mixed" with the FF causal
# conv as one half product) and "mixed" must keep a >= 4x margin
CEIL = {"exact": 1e-4, "mixed": 2.5e-4, "hybrid": 2.5e-4, "half": TOL}
_SWEEP_REF = {}                            # oracle outputs of the sweep, shared by the precision parametrisations
F_linear = torch.nn.functional.linear
from tests.parity_record import record  # noqa: E402  (key-wise merge into the tracked record; never a whole-file overwrite)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


def rel_rows(a, b):
    a, b = a.double().cpu().flatten(1), b.double().cpu().flatten(1)
    return ((a - b).norm(dim=1) / b.norm(dim=1)).tolist()


def gpu_oracle_ddim(sd, noise, steps, **kw):
    """the oracle's sampling loop (oracle/ns2_oracle.py: plain PyTorch fp32 ops) on GPU tensors -- the CPU loop of a 30-50 step trajectory was
    half a minute of a shared host per test -- pinned to the CPU oracle on one forward of the first utterance (< 5e-6)"""
    sdg = {k: v.to(DEV) for k, v in sd.items()}
    dev = lambda v: v.to(DEV) if torch.is_tensor(v) else v      # noqa: E731
    with torch.no_grad():
        t1 = torch.full((1,), 0.5)
        fkw = {k: v[:1] for k, v in kw.items() if torch.is_tensor(v)}
        if fkw:
            pin = rel(O.model_forward_with_cond_scale(sdg, noise[:1].to(DEV), t1.to(DEV), fkw.get("prompt").to(DEV), fkw.get("cond").to(DEV), kw.get("cond_scale", 1.0)),
                      O.model_forward_with_cond_scale(sd, noise[:1], t1, fkw.get("prompt"), fkw.get("cond"), kw.get("cond_scale", 1.0)))
        else:
            pin = rel(O.model_forward(sdg, noise[:1].to(DEV), t1.to(DEV)), O.model_forward(sd, noise[:1], t1))
        assert pin < 5e-6, f"GPU-resident oracle drifted from the CPU oracle: {pin}"
        return O.ddim_sample(sdg, noise.to(DEV), steps, **{k: dev(v) for k, v in kw.items()})


def build(kw, seed=1, precision="exact", scale_weights=1.0):
    m = Model(**kw, precision=precision)
    own = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = make_weights(own, seed=seed)
    if scale_weights != 1.0:
        sd = {k: (v * scale_weights if k.endswith("weight") and v.ndim >= 2 else v) for k, v in sd.items()}
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


# ------------------------------------------------------------------------------------------ mixed mode vs the reference
import glob  # noqa: E402


@pytest.mark.parametrize("precision", ["mixed", "hybrid"])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "model_*.pt"))), ids=os.path.basename)
def test_mixed_mode_matches_reference_golden(path, precision):
    """precision="mixed" (IEEE-half product + both first-order correction terms on the fp8 MFMA) and "hybrid" (the same with
    the FF causal conv as one half product) against the reference's own outputs, every golden configuration and guidance scale."""
    fix = torch.load(path, weights_only=False)
    kw, b, n = fix["kwargs"], fix["batch"], fix["n"]
    m = Model(**kw, precision=precision)
    m.load_state_dict(make_weights(fix["shapes"], seed=fix["weight_seed"]))
    m = m.to(DEV).eval()
    x = make_input("x", (b, n, kw["dim"]), seed=fix["input_seed"]).to(DEV)
    t = make_input("times", (b,), seed=fix["input_seed"], uniform=True).to(DEV)
    kws = {}
    if kw.get("condition_on_prompt"):
        kws = dict(prompt=make_input("prompt", (b, fix["n_prompt"], kw["dim_prompt"]), seed=fix["input_seed"]).to(DEV),
                   cond=make_input("cond", (b, kw["dim_prompt"], fix["n_cond"]), seed=fix["input_seed"]).to(DEV))
    worst = 0.0
    with torch.no_grad():
        for name, ref in fix["outputs"].items():
            y = m.forward_with_cond_scale(x, t, cond_scale=float(name.split("_")[-1]), **kws)
            assert torch.isfinite(y).all()
            worst = max(worst, rel(y, ref))
    record(f"{precision}_golden/{os.path.basename(path)}", worst)
    assert 1e-7 < worst < CEIL[precision], f"{precision}-mode rel {worst}"


def test_precision_sweep_headline_architecture():
    """d512/L12 x 1024 frames: 8 weight seeds x diffusion times {0.002, 0.5, 0.999} (one utterance each), per-utterance error
    against the fp32 oracle; the MAX over the sweep is what is asserted and what bench.py quotes.  Round 6: one model per seed serves
    the three arithmetics (`model.precision = ...` re-packs the same parameters) and the oracle runs on GPU tensors, pinned to the CPU
    oracle on the first seed -- the sweep took 207 s of the suite's 733 s."""
    kw = dict(dim=512, depth=12)
    times = torch.tensor([0.002, 0.5, 0.999])
    errs = {p: [] for p in ("hybrid", "mixed", "half")}
    for seed in range(8):
        m, sd = build(kw, seed=100 + seed, precision="hybrid")
        x = make_input("x", (3, 1024, 512), seed=200 + seed)
        with torch.no_grad():
            ref = O.model_forward({k: v.to(DEV) for k, v in sd.items()}, x.to(DEV), times.to(DEV))
            if seed == 0:
                pin = rel(ref[:1], O.model_forward(sd, x[:1], times[:1]))
                assert pin < 5e-6, f"GPU-resident oracle drifted from the CPU oracle: {pin}"
            for precision in errs:
                m.precision = precision
                y = m(x.to(DEV), times.to(DEV))
                assert torch.isfinite(y).all()
                errs[precision].append(rel_rows(y, ref))
        del m, ref
        torch.cuda.empty_cache()
    for precision, e in errs.items():
        flat = [v for row in e for v in row]
        record(f"sweep_d512_L12/{precision}", dict(max=max(flat), mean=sum(flat) / len(flat), per_seed_per_time=e,
                                                   times=times.tolist(), seeds=8))
        print(f"{precision}: max {max(flat):.2e} mean {sum(flat) / len(flat):.2e} over 8 seeds x 3 times")
    for precision, e in errs.items():
        assert max(v for row in e for v in row) < CEIL[precision], f"{precision}: max rel err over the sweep {max(v for row in e for v in row)}"


_COND_REF = {}


def test_conditioned_cfg_d512():
    """BASELINE config 3 architecture with classifier-free guidance (two forwards mixed at cond_scale 1.3, NS2:914-927),
    three weight / input seeds, three arithmetics on the same model object; the maximum is recorded and asserted."""
    kw = dict(dim=512, depth=12, dim_prompt=512, condition_on_prompt=True)
    b, n = 2, 512
    errs = {p: [] for p in ("hybrid", "mixed", "half")}
    for seed in (9, 19, 29):
        m, sd = build(kw, seed=seed, precision="hybrid")
        x = make_input("x", (b, n, 512), seed=seed + 1)
        t = make_input("times", (b,), seed=seed + 1, uniform=True)
        prompt = make_input("prompt", (b, 103, 512), seed=seed + 1)
        cond = make_input("cond", (b, 512, n), seed=seed + 1)
        with torch.no_grad():
            ref = O.model_forward_with_cond_scale({k: v.to(DEV) for k, v in sd.items()}, x.to(DEV), t.to(DEV), prompt.to(DEV), cond.to(DEV), 1.3)
            if seed == 9:
                pin = rel(ref[:1], O.model_forward_with_cond_scale(sd, x[:1], t[:1], prompt[:1], cond[:1], 1.3))
                assert pin < 5e-6, f"GPU-resident oracle drifted from the CPU oracle: {pin}"
            for precision in errs:
                m.precision = precision
                y = m.forward_with_cond_scale(x.to(DEV), t.to(DEV), prompt=prompt.to(DEV), cond=cond.to(DEV), cond_scale=1.3)
                assert torch.isfinite(y).all()
                errs[precision].append(rel(y, ref))
        del m, ref
        torch.cuda.empty_cache()
    for precision, e in errs.items():
        record(f"conditioned_cfg_d512/{precision}", max(e))
        record(f"conditioned_cfg_d512_per_seed/{precision}", e)
    for precision, e in errs.items():
        assert max(e) < CEIL[precision], f"{precision}: rel {e}"


def test_ddim_trajectory_50_steps():
    """error compounding along a sampling trajectory: 50 DDIM steps at d128/L6 (timesteps=50), all precisions, against the
    oracle's loop on the same injected noise"""
    kw = dict(dim=128, depth=6)
    noise = make_input("noise", (2, 256, 128), seed=31)
    out = {}
    ref = None
    for precision in ("exact", "mixed", "hybrid", "half"):
        m, sd = build(kw, seed=30, precision=precision)
        d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=50)
        y = d.sample(length=256, batch_size=2, noise=noise)
        if ref is None:
            ref = gpu_oracle_ddim(sd, noise, 50)
        assert torch.isfinite(y).all()
        out[precision] = rel(y, ref)
    record("ddim_50_steps_d128_L6", out)
    print("50-step DDIM trajectory rel err:", {k: f"{v:.2e}" for k, v in out.items()})
    assert out["exact"] < 1e-4 and out["mixed"] < 5e-4 and out["hybrid"] < 5e-4 and out["half"] < 5e-3, out


def test_ddim_trajectory_headline_architecture():
    """10 DDIM steps at the headline architecture (d512/L12, one utterance of 512 frames) in the benched plan and its neighbours"""
    kw = dict(dim=512, depth=12)
    noise = make_input("noise", (1, 512, 512), seed=38)
    out = {}
    m, sd = build(kw, seed=37, precision="hybrid")             # one model object, three arithmetics (the packs are per precision)
    ref = gpu_oracle_ddim(sd, noise, 10)
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=10)
    for precision in ("hybrid", "mixed", "half"):
        m.precision = precision
        y = d.sample(length=512, batch_size=1, noise=noise)
        assert torch.isfinite(y).all()
        out[precision] = rel(y, ref)
    del m, d
    torch.cuda.empty_cache()
    record("ddim_10_steps_d512_L12", out)
    print("10-step DDIM trajectory at d512/L12 rel err:", {k: f"{v:.2e}" for k, v in out.items()})
    assert out["hybrid"] < CEIL["hybrid"] and out["mixed"] < CEIL["mixed"] and out["half"] < 2e-3, out


def test_conditioned_ddim_trajectory_with_cfg():
    """a conditioned sampling run with classifier-free guidance: 30 DDIM steps at d128/L6 (dim_prompt 128, prompt of 60 encoded
    frames, frame-aligned cond, cond_scale 1.3): two forwards per step, the step-invariant conditioning computed once -- against
    the oracle's loop on the same injected noise, every benched precision"""
    kw = dict(dim=128, depth=6, dim_prompt=128, condition_on_prompt=True)
    noise = make_input("noise", (2, 256, 128), seed=36)
    p_enc = make_input("prompt_enc", (2, 60, 128), seed=36)
    cond = make_input("cond", (2, 128, 256), seed=36)
    out, ref = {}, None
    for precision in ("exact", "mixed", "hybrid", "half"):
        m, sd = build(kw, seed=35, precision=precision)
        d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=30).to(DEV).eval()
        y = d.sample(length=256, prompt_enc=p_enc.to(DEV), cond=cond.to(DEV), cond_scale=1.3, noise=noise)
        if ref is None:
            ref = gpu_oracle_ddim(sd, noise, 30, prompt=p_enc, cond=cond, cond_scale=1.3)
        assert torch.isfinite(y).all()
        out[precision] = rel(y, ref)
    record("ddim_30_steps_conditioned_cfg_d128_L6", out)
    print("conditioned 30-step DDIM trajectory rel err:", {k: f"{v:.2e}" for k, v in out.items()})
    assert out["exact"] < 1e-4 and out["mixed"] < 5e-4 and out["hybrid"] < 5e-4 and out["half"] < 5e-3, out


def test_large_activation_stress():
    """weights x8 (trained checkpoints have larger activations than N(0, 1/fan_in) init): outputs stay finite in every mode
    (IEEE-half conversions saturate at +-65504 / +-57344 instead of producing inf) and the error is reported"""
    kw = dict(dim=128, depth=6)
    x = make_input("x", (2, 256, 128), seed=41) * 4.0
    t = torch.tensor([0.3, 0.9])
    out = {}
    for scale in (2.0, 8.0):
        ref = None
        for precision in ("exact", "mixed", "hybrid", "half"):
            m, sd = build(kw, seed=40, precision=precision, scale_weights=scale)
            with torch.no_grad():
                y = m(x.to(DEV), t.to(DEV))
                if ref is None:
                    ref = O.model_forward(sd, x, t)          # (the weights depend on the scale only)
            assert torch.isfinite(y).all(), f"{precision} x{scale}: non-finite output"
            out[f"x{scale:g}/{precision}"] = rel(y, ref)
    record("large_activation_stress_d128_L6", out)
    print("stress rel err:", {k: f"{v:.2e}" for k, v in out.items()})
    assert out["x2/exact"] < 1e-4 and out["x2/mixed"] < CEIL["mixed"] and out["x2/hybrid"] < CEIL["hybrid"] and out["x2/half"] < 2e-3, out
    assert out["x8/exact"] < 1e-3, out
    # weights x8 push activations past the IEEE-half range: precisions "half" / "mixed" clamp (finite, wrong by O(1), see the
    # numbers above) -- the range guard must say so loudly instead of returning the clamped result
    from naturalspeech2_pytorch_amd import Ns2Error
    noise = make_input("noise", (2, 256, 128), seed=42) * 4.0
    for precision in ("hybrid", "mixed", "half"):
        ops.saturation_count(reset=True)
        ok_model, _ = build(kw, seed=40, precision=precision, scale_weights=2.0)
        d = NaturalSpeech2(ok_model, codec=None, target_sample_hz=24000, timesteps=2)
        d.sample(length=256, batch_size=2, noise=noise)                       # in range: no complaint
        bad_model, sd_bad = build(kw, seed=40, precision=precision, scale_weights=8.0)
        d = NaturalSpeech2(bad_model, codec=None, target_sample_hz=24000, timesteps=2)
        with pytest.raises(Ns2Error, match="IEEE-half range"):
            d.ddim_sample((2, 256, 128), noise=noise, on_saturation="raise")
        # round 4, the default: the run is repeated with precision="exact" and the model stays there -- a fast mode survives a
        # checkpoint whose activations leave the half range, and returns the right audio
        bad_model, sd_bad = build(kw, seed=40, precision=precision, scale_weights=8.0)
        d = NaturalSpeech2(bad_model, codec=None, target_sample_hz=24000, timesteps=2)
        with pytest.warns(UserWarning, match="repeating the sampling run"):
            got = d.sample(length=256, batch_size=2, noise=noise)
        assert bad_model.precision == "exact"
        e = rel(got, O.ddim_sample(sd_bad, noise, 2))
        out[f"x8/{precision}_sampler_demoted"] = e
        assert e < 1e-3, (precision, e)
    record("large_activation_stress_d128_L6", out)
    ex_model, _ = build(kw, seed=40, precision="exact", scale_weights=8.0)
    NaturalSpeech2(ex_model, codec=None, target_sample_hz=24000, timesteps=2).sample(length=256, batch_size=2, noise=noise)


def test_hybrid_plan_with_amplified_ff_branch():
    """The hybrid plan drops the correction terms of the FF causal conv and of the Wavenet's dilated convs.  At random init the FF
    branch is dominated by its biases and hides its own rounding, so the plan is also checked with every FF-in weight x6 (GEGLU
    output x36: the branch's data term dominates, like a trained branch that carries signal): it must keep a >= 2x margin there
    (tools/precision_study.py --scale-ffin 6 predicts 2.3e-4 for hybrid, 4.8e-5 for mixed, 7.2e-4 for half)."""
    kw = dict(dim=128, depth=6)
    x = make_input("x", (2, 256, 128), seed=51)
    t = torch.tensor([0.2, 0.8])
    out = {}
    for precision in ("mixed", "hybrid", "half"):
        m = Model(**kw, precision=precision)
        sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=50)
        for k, v in sd.items():
            if v.ndim == 2 and v.shape[1] == 128 and v.shape[0] != 128 and v.shape[0] % 64 != 0:      # FF-in [2 * inner, dim]
                sd[k] = v * 6.0
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        with torch.no_grad():
            y = m(x.to(DEV), t.to(DEV))
            ref = O.model_forward(sd, x, t)
        out[precision] = rel(y, ref)
    record("ff_branch_x6_d128_L6", out)
    print("FF-in x6:", {k: f"{v:.2e}" for k, v in out.items()})
    assert out["mixed"] < 2.5e-4 and out["hybrid"] < 5e-4 and out["half"] < 2e-3, out


def test_precision_plans_with_outlier_channels_and_heavy_tails():
    """the tolerance had only been shown on random-init weights.  No trained checkpoint can be had here (no network), so the
    statistics trained transformers are known for are imposed on the headline architecture instead: (a) OUTLIER CHANNELS -- in every weight matrix
    four input columns x16 and four output rows x8 (residual-stream channels two orders of magnitude above the rest, the case that breaks narrow
    formats); (b) HEAVY TAILS -- every weight multiplied elementwise by exp(0.8 z), z ~ N(0, 1) (kurtosis ~ 30).  d512 / L12, 2 x 1024 frames,
    every plan against the fp32 oracle (GPU-resident, pinned to the CPU oracle on one utterance).  Recorded; asserted: exact / mixed / hybrid stay
    inside the 1e-3 tolerance, half is printed (it is the plan without correction terms)."""
    kw = dict(dim=512, depth=12)
    b, n = 2, 1024
    x = make_input("x", (b, n, 512), seed=141)
    t = torch.tensor([0.3, 0.9])
    base = make_weights({k: tuple(v.shape) for k, v in Model(**kw).state_dict().items()}, seed=140)
    g = torch.Generator().manual_seed(142)

    def outliers(sd):
        out = {}
        for k, v in sd.items():
            v = v.clone()
            if v.ndim >= 2 and min(v.shape[0], v.shape[1]) >= 64:
                cols = torch.randperm(v.shape[1], generator=g)[:4]
                rows = torch.randperm(v.shape[0], generator=g)[:4]
                v[:, cols] *= 16.0
                v[rows] *= 8.0
            out[k] = v
        return out

    def heavy(sd):
        return {k: (v * torch.exp(0.8 * torch.randn(v.shape, generator=g)) if v.ndim >= 2 else v.clone()) for k, v in sd.items()}

    res = {}
    for name, make in (("outlier_channels", outliers), ("heavy_tails", heavy)):
        sd = make(base)
        sdg = {k: v.to(DEV) for k, v in sd.items()}
        with torch.no_grad():
            ref = O.model_forward(sdg, x.to(DEV), t.to(DEV))
            pin = rel(ref[:1], O.model_forward(sd, x[:1], t[:1]))
        assert pin < 5e-6 and torch.isfinite(ref).all(), pin
        res[name] = {"oracle_output_rms": float(ref.pow(2).mean().sqrt())}
        m = Model(**kw, precision="exact")                 # one model object per weight set, four arithmetics (the packs are per precision)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        for precision in ("exact", "mixed", "hybrid", "half"):
            m.precision = precision
            with torch.no_grad():
                y = m(x.to(DEV), t.to(DEV))
            m.check_saturation(sync=True)
            res[name][precision] = rel(y, ref)
        del m
        print(name, {k: f"{v:.2e}" for k, v in res[name].items()})
        assert res[name]["exact"] < 1e-4 and res[name]["mixed"] < 1e-3 and res[name]["hybrid"] < 1e-3, res[name]
    record("stress_d512_L12_b2x1024", res)


def test_half_conversion_saturates():
    x = torch.tensor([[1e6, -1e6, 65504.0, 7e4] + [0.0] * 28], device=DEV)
    h = ops.join(ops.split(x, precision=2))[0, :4].tolist()
    assert h == [65504.0, -65504.0, 65504.0, 65504.0]
    m = ops.join(ops.split(x, precision=4))[0, :4]
    assert torch.isfinite(m).all() and m[0].item() == 57344.0 and m[1].item() == -57344.0


# ------------------------------------------------------------------------------------------ mixed-mode plane format
def test_h8_plane_format_contract():
    """include/ns2hip.h: precision-4 operands are 128-byte lines [half x 32 | e5m2(x) x 32 | e5m2((x - half(x)) * 2^12) x 32]"""
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(70, 100, generator=g) * torch.logspace(-3, 3, 100)[None]).to(DEV)
    p = ops.split(x, precision=4)
    assert (p.fmt, p.rows, p.ld) == ("h8", 70, 128) and p.buf.shape == (70, 256) and p.buf.dtype == torch.float16
    xc = x.cpu()
    hi = xc.to(torch.float16)
    assert torch.equal(p.hi_plane()[:, :100].cpu(), hi)
    h8, l8 = p.byte_planes()
    e_h8 = xc.to(torch.float8_e5m2).view(torch.uint8)
    e_l8 = ((xc - hi.float()) * 4096.0).to(torch.float8_e5m2).view(torch.uint8)
    assert torch.equal(h8[:, :100].cpu(), e_h8)                     # hardware v_cvt_pk_bf8_f32 == IEEE RNE e5m2
    assert torch.equal(l8[:, :100].cpu(), e_l8)
    assert p.buf.reshape(70, 4, 2, 32)[:, 3, 0, 4:].abs().sum().item() == 0            # zero padding (columns 100..127)
    y = ops.join(p, 100)
    assert rel(y, x) < 2 ** -13                                     # half + l8 * 2^-12 restores ~14 significand bits


# ------------------------------------------------------------------------------------------ cache invalidation
def test_param_data_mutation_triggers_repack():
    """ema_pytorch writes the shadow model through `.data` (no version bump): `refresh_weights()` (called by sample()) must
    notice by content and re-pack."""
    m, sd = build(dict(dim=64, depth=1), seed=13)
    x = make_input("x", (1, 40, 64), seed=14).to(DEV)
    t = torch.tensor([0.3], device=DEV)
    with torch.no_grad():
        y0 = m(x, t)
        w = getattr(m.transformer.layers[0], "1").to_q.weight
        v0 = w._version
        w.data.mul_(1.5)                                            # in-place through .data: version counter unchanged
        assert w._version == v0
        assert torch.equal(m(x, t), y0)                             # cheap signature cannot see it (documented)
        assert m.refresh_weights() is True                          # content fingerprint does
        y1 = m(x, t)
        assert m.refresh_weights() is False
    sd2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    assert not torch.equal(y1, y0) and rel(y1, O.model_forward(sd2, x.cpu(), t.cpu())) < 1e-4
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=2)
    n = make_input("noise", (1, 40, 64), seed=15)
    a = d.sample(length=40, noise=n)
    getattr(m.transformer.layers[0], "1").to_q.weight.data.mul_(0.5)
    b = d.sample(length=40, noise=n)                                # sample() refreshes by itself
    assert not torch.equal(a, b)


# ------------------------------------------------------------------------------------------ the reference's loop as caller
def test_reference_caller_loop_drives_hip_model():
    """NS2:1379-1431 restated as a caller (oracle.reference_caller_ddim_sample: per step `model.forward_with_cond_scale(audio,
    times, prompt=, cond_scale=, cond=)` + the reference's unfused update chain in torch) around the HIP Model, against the
    golden trajectory produced by the unmodified reference; and the fused sampler of this package against the same."""
    fix = torch.load(os.path.join(GOLD, "ddim_uncond_d64.pt"), weights_only=False)
    m = Model(**fix["kwargs"])
    m.load_state_dict(make_weights(fix["shapes"], seed=fix["weight_seed"]))
    m = m.to(DEV).eval()
    noise = make_input("noise", (fix["batch"], fix["n"], fix["kwargs"]["dim"]), seed=fix["input_seed"])
    out = O.reference_caller_ddim_sample(m, noise.to(DEV), fix["timesteps"])
    assert rel(out, fix["output"]) < 5e-4
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=fix["timesteps"])
    assert rel(d.sample(length=fix["n"], batch_size=fix["batch"], noise=noise), out) < 1e-5


@pytest.mark.parametrize("schedule", ["sigmoid", "cosine", "linear"])
@pytest.mark.parametrize("objective", ["v", "eps", "x0"])
def test_sampler_schedules_and_objectives(schedule, objective):
    """NS2:1133-1148 schedules x NS2:1414-1422 objectives through the fused HIP update (cosine is unrunnable upstream:
    restated intent, parity unpinned)"""
    m, sd = build(dict(dim=64, depth=1), seed=50)
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=4, noise_schedule=schedule, objective=objective)
    noise = make_input("noise", (2, 48, 64), seed=51)
    y = d.sample(length=48, batch_size=2, noise=noise)
    with torch.no_grad():
        ref = O.ddim_sample(sd, noise, 4, objective=objective, schedule=schedule)
    assert rel(y, ref) < 5e-4, (schedule, objective, rel(y, ref))
    # non-default schedule parameters go through the host-side schedule functions (same model steps)
    d2 = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=4, noise_schedule="sigmoid", objective=objective,
                        schedule_kwargs=dict(start=-3, end=3, tau=1))
    if schedule == "sigmoid":
        assert rel(d2.sample(length=48, batch_size=2, noise=noise), ref) < 5e-4


# ------------------------------------------------------------------------------------------ no shared mutable state
def test_two_models_on_two_streams_do_not_interfere():
    """the library keeps no mutable host state and no library-owned scratch: two models driven from two
    streams, interleaved, give the results of running each alone.  d=512 takes the split-K conditioning path."""
    kw = dict(dim=512, depth=1, wavenet_layers=2, wavenet_stacks=1)
    ma, _ = build(kw, seed=60)
    mb, _ = build(kw, seed=61, precision="mixed")
    xa, xb = make_input("x", (32, 64, 512), seed=62).to(DEV), make_input("x", (32, 64, 512), seed=63).to(DEV)
    ta, tb = torch.rand(32, device=DEV), torch.rand(32, device=DEV)
    with torch.no_grad():
        ya, yb = ma(xa, ta), mb(xb, tb)
        torch.cuda.synchronize()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        outs_a, outs_b = [], []
        for _ in range(6):
            with torch.cuda.stream(sa):
                outs_a.append(ma(xa, ta))
            with torch.cuda.stream(sb):
                outs_b.append(mb(xb, tb))
        torch.cuda.synchronize()
    assert all(torch.equal(o, ya) for o in outs_a) and all(torch.equal(o, yb) for o in outs_b)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two devices")
def test_two_models_on_two_devices_one_process():
    kw = dict(dim=64, depth=2)
    outs = []
    for i in range(2):
        dev = torch.device("cuda", i)
        m = Model(**kw)
        sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=70)
        m.load_state_dict(sd)
        m = m.to(dev).eval()
        x = make_input("x", (2, 96, 64), seed=71)
        t = make_input("times", (2,), seed=71, uniform=True)
        with torch.cuda.device(dev), torch.no_grad():
            outs.append(m(x.to(dev), t.to(dev)).cpu())
    ref = O.model_forward(sd, x, t)
    assert rel(outs[0], ref) < 1e-4 and torch.equal(outs[0], outs[1])


def test_sharded_sample_over_rccl_world1():
    """the data-parallel sampler with the real Model over an RCCL ("nccl") process group; world 1 on a one-GPU box, so the
    collective degenerates but the whole code path (init, shard, all_gather, destroy) is the one the 8-GPU run takes"""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29700 + os.getpid() % 200)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        m, _ = build(dict(dim=64, depth=1), seed=18)
        d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=3)
        fn = lambda noise: d.ddim_sample(tuple(noise.shape), noise=noise)   # noqa: E731
        out = D.sharded_sample(fn, total=5, length=32, dim=64, seed=9, device=DEV)
        bufs = [torch.empty_like(out)]
        dist.all_gather(bufs, out)                                   # RCCL really runs
        direct = fn(D.utterance_noise(0, 5, 32, 64, seed=9, device=DEV))
        assert torch.equal(bufs[0], direct) and out.shape == (5, 32, 64)
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------ RVQ: test what is claimed
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_rvq_config4_size_every_mismatch_adjudicated(seed):
    """BASELINE config 4 size (32 x 1024 frames x 8 codebooks x 1024 codes).  The claim is: indices equal to the reference
    formula wherever that formula is well-defined.  `dist = -(|x|^2 - 2 x E^T + |E|^2)` evaluated in fp32 depends on the
    BLAS summation order, so the oracle is run twice, in fp32 and in fp64: every row must equal the fp32 oracle, or -- at
    its FIRST differing stage -- the fp64 oracle (i.e. the fp32 oracle itself is not the nearest code there)."""
    cb = make_input("codebooks", (8, 1024, 128), seed=300 + seed)
    x = make_input("latents", (32 * 1024, 128), seed=400 + seed)
    codes, emb, ties = ops.rvq_encode(x.to(DEV), cb.to(DEV), count_ties=True)
    codes = codes.cpu()
    c32, e32, _ = R.rvq_encode(x, cb)
    mism = (codes != c32).any(dim=-1).nonzero().flatten()
    n_unexplained = 0
    if len(mism):
        c64, _, _ = R.rvq_encode(x[mism].double(), cb.double())
        for i, r in enumerate(mism.tolist()):
            q = int((codes[r] != c32[r]).float().argmax())
            if not torch.equal(codes[r, : q + 1], c64[i, : q + 1]):
                n_unexplained += 1
    record(f"rvq_config4/seed{seed}", dict(rows=x.shape[0], rows_differing_from_fp32_oracle=len(mism),
                                           unexplained=n_unexplained, near_tie_redecisions=int(ties.item())))
    assert n_unexplained == 0, f"{n_unexplained} rows differ from BOTH the fp32 and the fp64 oracle"
    same = torch.ones(x.shape[0], dtype=torch.bool)
    same[mism] = False
    assert torch.equal(emb.cpu()[same], e32[same])                  # summed embeddings bit-equal where the indices agree


def _hf_encodec():
    tf = pytest.importorskip("transformers")
    torch.manual_seed(0)
    model = tf.EncodecModel(tf.EncodecConfig()).eval()
    g = torch.Generator().manual_seed(1)
    for layer in model.quantizer.layers:                            # HF zero-initialises the codebooks (HFENC:355)
        layer.codebook.embed.copy_(torch.randn(layer.codebook.embed.shape, generator=g))
    return model.to(DEV)


@pytest.mark.parametrize("precision,tol", [("exact", 2e-4), ("mixed", 1e-3)])
def test_seanet_encoder_decoder_match_hf(precision, tol):
    """EnCodec's SEANet encoder (strided causal convs with reflect padding, ResnetBlocks, 2-layer LSTM) and
    decoder (transposed convs) on the HIP kernels against HF's own modules (HFENC:285-347) on the same random-init weights"""
    from naturalspeech2_pytorch_amd import SEANetDecoderHIP, SEANetEncoderHIP
    hf = _hf_encodec()
    wav = make_input("wav", (2, 1, 320 * 50), seed=92).to(DEV)
    with torch.no_grad():
        ref_lat = hf.encoder(wav)                                   # [2, 128, 50]
        lat = SEANetEncoderHIP(hf.encoder, precision=precision)(wav)
        assert lat.shape == ref_lat.shape == (2, 128, 50)
        e_enc = rel(lat, ref_lat)
        ref_wav = hf.decoder(ref_lat)                               # [2, 1, 16000]
        out = SEANetDecoderHIP(hf.decoder, precision=precision)(ref_lat)
        assert out.shape == ref_wav.shape == (2, 1, 320 * 50)
        e_dec = rel(out, ref_wav)
    record(f"seanet_vs_hf/{precision}", dict(encoder=e_enc, decoder=e_dec))
    print(f"SEANet {precision}: encoder rel {e_enc:.2e}, decoder rel {e_dec:.2e}")
    assert e_enc < tol and e_dec < tol, (e_enc, e_dec)


def test_seanet_encoder_batch_beyond_one_lstm_launch():
    """40 utterances: the one-launch LSTM recurrences take at most 32 batch rows, larger batches go through them in chunks (rows are
    independent through the recurrence) -- against HF's encoder"""
    from naturalspeech2_pytorch_amd import SEANetEncoderHIP
    hf = _hf_encodec()
    wav = make_input("wav40", (40, 1, 320 * 12), seed=98).to(DEV)
    with torch.no_grad():
        ref_lat = hf.encoder(wav)
        lat = SEANetEncoderHIP(hf.encoder)(wav)
    assert lat.shape == ref_lat.shape == (40, 128, 12)
    assert rel(lat, ref_lat) < 2e-4 and rel(lat[32:], ref_lat[32:]) < 2e-4


def test_seanet_pieces():
    """the pieces around the GEMMs: reflect prefix + ELU + im2col (ns2_seanet_prep), and one LSTM layer vs torch.nn.LSTM"""
    from naturalspeech2_pytorch_amd import seanet as S
    import torch.nn.functional as F
    x = make_input("x", (2 * 40, 8), seed=93).to(DEV)
    pl = S._prep(x, 2, 40, 8, elu=True, prefix=3, precision=3)
    got = ops.join(pl, 8).reshape(2, 43, 8)
    xe = F.elu(x.reshape(2, 40, 8))
    assert rel(got[:, 3:], xe) < 1e-5 and rel(got[:, :3], xe[:, 1:4].flip(1)) < 1e-5          # row -j mirrors row j
    x1 = make_input("x1", (2 * 40, 1), seed=94).to(DEV)
    im = ops.join(S._prep(x1, 2, 40, 1, im2col_k=7, precision=3), 7).reshape(2, 40, 7)
    xp = F.pad(x1.reshape(2, 1, 40), (6, 0), mode="reflect")[:, 0]
    assert rel(im, xp.unfold(1, 7, 1)) < 1e-5
    lstm = torch.nn.LSTM(64, 64, 1).to(DEV)
    xin = make_input("xl", (3, 20, 64), seed=95).to(DEV)
    with torch.no_grad():
        ref = lstm(xin.transpose(0, 1))[0].transpose(0, 1) + xin
        xproj = F.linear(xin.reshape(60, 64), lstm.weight_ih_l0, lstm.bias_ih_l0).contiguous()
        state = torch.empty(3 * 3 * 64, device=DEV)
        out = torch.empty(60, 64, device=DEV)
        from naturalspeech2_pytorch_amd import _lib
        _lib.check(_lib.load().ns2_lstm_layer(xproj.data_ptr(), 256, lstm.weight_hh_l0.data_ptr(), lstm.bias_hh_l0.data_ptr(),
                                              state.data_ptr(), state.numel(), xin.reshape(60, 64).data_ptr(), 64, out.data_ptr(), 64, 3, 20, 64,
                                              torch.cuda.current_stream().cuda_stream), "lstm")
    assert rel(out.reshape(3, 20, 64), ref) < 1e-5


@pytest.mark.parametrize("B,T", [(1, 37), (5, 37), (8, 37), (19, 37), (32, 37), (32, 700), (3, 1024)])
def test_lstm_persistent_recurrence(B, T):
    """EnCodec's LSTM width (H = 512): the one-launch persistent recurrence (steps synchronised through tagged {h, step} pairs)
    against torch.nn.LSTM and against the per-step kernel (the same entry point with only the minimal scratch), ragged batch sizes
    across the 8-row groups, and sequences long enough that every exchange buffer is reused hundreds of times"""
    from naturalspeech2_pytorch_amd import _lib
    lib = _lib.load()
    H = 512
    torch.manual_seed(B)
    lstm = torch.nn.LSTM(H, H, 1).to(DEV)
    xin = make_input("xl", (B, T, H), seed=96 + B).to(DEV)
    with torch.no_grad():
        ref = lstm(xin.transpose(0, 1))[0].transpose(0, 1) + xin
        xproj = F_linear(xin.reshape(B * T, H), lstm.weight_ih_l0, lstm.bias_ih_l0).contiguous()
    outs = []
    for nstate in (int(lib.ns2_lstm_state_floats(B, H)), 3 * B * H):
        state = torch.empty(nstate, device=DEV)
        for _ in range(2):                                           # the second launch finds the first one's tags in the scratch
            out = torch.full((B * T, H), float("nan"), device=DEV)
            _lib.check(lib.ns2_lstm_layer(xproj.data_ptr(), 4 * H, lstm.weight_hh_l0.data_ptr(), lstm.bias_hh_l0.data_ptr(), state.data_ptr(),
                                          nstate, xin.reshape(B * T, H).data_ptr(), H, out.data_ptr(), H, B, T, H,
                                          torch.cuda.current_stream().cuda_stream), "lstm")
            torch.cuda.synchronize()
        outs.append(out.reshape(B, T, H))
        assert rel(outs[-1], ref) < 1e-5, f"scratch {nstate}: {rel(outs[-1], ref)}"
    if int(lib.ns2_lstm_state_floats(B, H)) > 3 * B * H:            # the two calls really took different kernels
        assert rel(outs[0], outs[1]) < 1e-5


@pytest.mark.parametrize("B,T", [(3, 40), (8, 33), (19, 150), (32, 700), (32, 31)])
def test_lstm2_both_layers_in_one_launch(B, T):
    """ns2_lstm2: EnCodec's 2-layer LSTM with layer 2 overlapped one frame behind layer 1 (layer 1 forms layer 2's input
    projections, handed over through a ring of 32 tagged slots per lane with an acknowledge word) against torch.nn.LSTM(num_layers=2)
    + skip; sequences shorter and much longer than the ring, ragged batch groups, and a second launch on the same scratch"""
    from naturalspeech2_pytorch_amd import _lib
    lib = _lib.load()
    H = 512
    torch.manual_seed(100 + B)
    lstm = torch.nn.LSTM(H, H, 2).to(DEV)
    xin = make_input("xl2", (B, T, H), seed=196 + B).to(DEV)
    with torch.no_grad():
        ref = lstm(xin.transpose(0, 1))[0].transpose(0, 1) + xin
        xproj = F_linear(xin.reshape(B * T, H), lstm.weight_ih_l0, lstm.bias_ih_l0).contiguous()
    nstate = int(lib.ns2_lstm2_state_floats())
    state = torch.randn(nstate, device=DEV)                            # garbage in the scratch must not matter
    for _ in range(2):
        out = torch.full((B * T, H), float("nan"), device=DEV)
        rc = lib.ns2_lstm2(xproj.data_ptr(), 4 * H, lstm.weight_hh_l0.data_ptr(), lstm.bias_hh_l0.data_ptr(), lstm.weight_ih_l1.data_ptr(),
                           lstm.bias_ih_l1.data_ptr(), lstm.weight_hh_l1.data_ptr(), lstm.bias_hh_l1.data_ptr(), state.data_ptr(), nstate,
                           xin.reshape(B * T, H).data_ptr(), H, out.data_ptr(), H, B, T, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, f"ns2_lstm2 rc={rc} (1 = unavailable: a whole MI355X must be able to hold its workgroups)"
        torch.cuda.synchronize()
        assert rel(out.reshape(B, T, H), ref) < 1e-5, rel(out.reshape(B, T, H), ref)
    import ctypes
    n = ctypes.c_int64(0)
    _lib.check(lib.ns2_lstm_abort_count(1, ctypes.byref(n)), "abort count")
    assert n.value == 0


def test_seanet_lstm_falls_back_when_a_launch_gave_up():
    """a one-launch recurrence that could not get all its workgroups resident is reported by ns2_lstm_abort_count; seanet.py then
    discards its output and runs the recurrence the next way (one launch per layer, then per frame).  The give-up is injected
    through the test hook; the encoder output must not change"""
    from naturalspeech2_pytorch_amd import _lib
    from naturalspeech2_pytorch_amd.seanet import SEANetEncoderHIP
    hf = _hf_encodec()
    enc = SEANetEncoderHIP(hf.encoder).to(DEV).eval()
    wav = make_input("wavfb", (3, 1, 320 * 40), seed=97).to(DEV)
    with torch.no_grad():
        ref = enc(wav)
        _lib.check(_lib.load().ns2_debug_lstm_inject_abort(1), "inject")
        with pytest.warns(UserWarning, match="gave up waiting for a frame"):
            out = enc(wav)
    assert rel(out, ref) < 1e-5
    import ctypes
    n = ctypes.c_int64(-1)
    _lib.check(_lib.load().ns2_lstm_abort_count(0, ctypes.byref(n)), "count")
    assert n.value == 0


def test_encodec_wrapper_with_hf_seanet():
    """the boundary class `codec(x, return_encoded=True)` on raw audio (BASELINE configs 1 / 4 input shape randn(4, 327680)):
    HF EnCodec's SEANet encoder injected, RVQ in HIP, against HF's own quantizer on the same encoder output"""
    hf = _hf_encodec()
    codec = EncodecWrapperHIP.from_hf(hf, hip_seanet=False).to(DEV).eval()     # HF's SEANet modules: isolates the RVQ comparison
    wav = make_input("wav", (4, 327680), seed=90).to(DEV)
    with torch.no_grad():
        emb, codes, _ = codec(wav, return_encoded=True)
        lat = hf.encoder(wav[:, None])                               # [4, 128, 1024]
        ref_codes = hf.quantizer.encode(lat, bandwidth=6.0)          # [8, 4, 1024]   (HFENC:424-438)
        ref_emb = hf.quantizer.decode(ref_codes)                     # [4, 128, 1024] (HFENC:440-447)
    assert emb.shape == (4, 1024, 128) and codes.shape == (4, 1024, 8) and codes.dtype == torch.int64
    ref_codes = ref_codes.permute(1, 2, 0)
    bad = (codes != ref_codes).any(dim=-1)
    frac = bad.float().mean().item()
    record("encodec_wrapper_hf", dict(frames=int(bad.numel()), frames_differing=int(bad.sum())))
    if bad.any():                                                   # adjudicate against exact arithmetic like above
        rows = bad.flatten().nonzero().flatten()
        xr = lat.transpose(1, 2).reshape(-1, 128)[rows].double().cpu()
        c64, _, _ = R.rvq_encode(xr, codec.rvq.codebooks.double().cpu())
        mine, theirs = codes.reshape(-1, 8)[rows].cpu(), ref_codes.reshape(-1, 8)[rows].cpu()
        for i in range(len(rows)):
            q = int((mine[i] != theirs[i]).float().argmax())
            assert torch.equal(mine[i, : q + 1], c64[i, : q + 1]), "index differs from HF's and from exact arithmetic"
    assert frac < 1e-3
    good = ~bad
    assert torch.allclose(emb[good], ref_emb.transpose(1, 2)[good], atol=1e-5)
    # curtail_from_left keeps the LAST whole frames (NS2:1445) and decode returns a waveform
    emb2, _, _ = codec(wav[:, : 320 * 10 + 7], return_encoded=True, curtail_from_left=True)
    assert emb2.shape == (4, 10, 128)
    assert codec.decode(emb2).shape == (4, 1, 3200)
    # the whole codec on HIP (SEANet encoder + RVQ + SEANet decoder) at the BASELINE config 1 / 4 input shape
    codec_hip = EncodecWrapperHIP.from_hf(hf).to(DEV).eval()
    with torch.no_grad():
        emb_h, codes_h, _ = codec_hip(wav, return_encoded=True)
        wav_h = codec_hip.decode(emb_h)
        wav_ref = hf.decoder(emb_h.transpose(1, 2))
    same = (codes_h == codes).all(dim=-1).float().mean().item()
    record("encodec_wrapper_hip_seanet", dict(frames_with_identical_codes=same, latent_rel=rel(emb_h, emb), decode_rel=rel(wav_h, wav_ref)))
    assert same > 0.98 and wav_h.shape == (4, 1, 327680) and rel(wav_h, wav_ref) < 2e-4


def test_naturalspeech2_with_codec_composition():
    """BASELINE config 1 as a composition (README:33-70): NaturalSpeech2(Model(dim=128, depth=6), codec): loss on raw audio
    + backward, then sample() back to a waveform.  (timesteps reduced from 1000: the loop body is what test_ddim_* cover.)"""
    hf = _hf_encodec()
    codec = EncodecWrapperHIP.from_hf(hf).to(DEV)
    model = Model(dim=128, depth=6).to(DEV)
    d = NaturalSpeech2(model=model, codec=codec, timesteps=3)
    raw = make_input("wav", (2, 32000), seed=91).to(DEV)
    loss = d(raw)
    loss.backward()
    assert torch.isfinite(loss) and model.wavenet.init_conv.weight.grad is not None
    d.eval()
    audio = d.sample(length=64)
    assert audio.shape == (1, 64 * 320) and torch.isfinite(audio).all()
    # the RVQ cross-entropy term of NS2:1670-1684
    d2 = NaturalSpeech2(model=model, codec=codec, timesteps=3, rvq_cross_entropy_loss_weight=0.1)
    l2 = d2(raw)
    assert torch.isfinite(l2) and l2.item() != loss.item()


# ------------------------------------------------------------------------------------------ bench.py is driver-runnable
def _run_bench(extra, env=None):
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--frames", "128",
           "--dim", "64", "--depth", "1", "--no-secondary", "--no-side", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_single_gpu():
    line = _run_bench(["--gpus", "1"])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["launches"] == 2 * 1      # 2 steps x 1 FF conv
    assert line["config"]["precision"] == "hybrid" and line["parity"]["live_rel_err_vs_fp32_oracle"]["hybrid"] < 2.5e-4
    line = _run_bench(["--gpus", "1", "--precision", "mixed"])
    assert line["roofline"]["launches"] == 2 * 3                                                       # + wavenet init conv + skip GEMM


def test_bench_self_spawns_ranks_without_a_launcher():
    """`python bench.py --gpus N` (no torchrun around it) must start N ranks itself.  On a one-GPU box
    the two ranks share the device, so the process group is gloo here; on the 8-GPU node it is nccl = RCCL."""
    line = _run_bench(["--gpus", "2"], env=dict(NS2_DIST_BACKEND="gloo"))
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["scaling"] == "weak"
    c = line["collective"]                     # the 8-GPU line explains itself
    assert c["backend"] == "gloo" and c["world"] == 2 and c["collectives_inside_the_loop"] == 0 and c["allgather_ms"] > 0
    assert c["allgather_bytes_per_rank"] == 2 * 128 * 64 * 4 and len(c["per_rank_ms_per_step"]["all"]) == 2
    assert 0 < c["per_rank_ms_per_step"]["min"] <= c["per_rank_ms_per_step"]["max"] <= line["ms_per_step"] * 1.001


def test_conditional_sample_with_hip_prompt_encoder():
    """NaturalSpeech2.sample on a conditional model with the reference keywords: `prompt` = codec latents [b, n_p, 128] goes
    through the HIP SpeechPromptEncoder (NS2:1474-1475), the aligned conditioning comes in as `cond` (the duration / pitch
    predictor is out of scope), classifier-free guidance at cond_scale 1.5; against the oracle's encoder + loop."""
    kw = dict(dim=64, depth=1, dim_prompt=128, condition_on_prompt=True)
    m, sd = build(kw, seed=80)
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=3, dim_codebook=64).to(DEV).eval()
    # a small prompt encoder (same class, reduced widths) so the CPU oracle stays cheap
    from naturalspeech2_pytorch_amd import SpeechPromptEncoder
    enc = SpeechPromptEncoder(64, dims=(96, 128), depth=1)
    esd = make_weights({k: tuple(v.shape) for k, v in enc.state_dict().items()}, seed=81)
    enc.load_state_dict(esd)
    d.prompt_enc = enc.to(DEV).eval()
    prompt = make_input("prompt", (2, 21, 64), seed=82)
    cond = make_input("cond", (2, 128, 40), seed=82)
    noise = make_input("noise", (2, 40, 64), seed=82)
    text = torch.randint(0, 100, (2, 12))
    y = d.sample(length=40, prompt=prompt.to(DEV), text=text.to(DEV), cond=cond.to(DEV), cond_scale=1.5, noise=noise)
    with torch.no_grad():
        p_enc = O.speech_prompt_encoder(prompt, esd, depth=1)
        ref = O.ddim_sample(sd, noise, 3, prompt=p_enc, cond=cond, cond_scale=1.5)
    assert y.shape == (2, 40, 64) and rel(y, ref) < 1e-3, rel(y, ref)
