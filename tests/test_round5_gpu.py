"""Round-5 GPU tests.

  * every gradient of the CONDITIONED model at the headline architecture (d512 / L12, 2 x 512 frames) against the
    reference's own autograd -- with the PerceiverResampler's backward now on the HIP Functions (training._resampler).
  * a frozen model with only x.requires_grad runs no weight-gradient GEMM and still returns the reference's dL/dx.
  (The helper loop of the tap-shared conv for half-empty column tiles -- gemm2.hip `helper` -- is covered by the shapes of
  tests/test_kernels_gpu.py::test_causal_conv whose last column tile has at most 128 valid columns: Cout = 300, 100, 341.)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from naturalspeech2_pytorch_amd import Model  # noqa: E402
from oracle.ref_stub import load_reference, reference_available  # noqa: E402
from tests.golden.gen import make_input, make_weights  # noqa: E402
from tests.parity_record import record  # noqa: E402

DEV = torch.device("cuda:0")
needs_ref = pytest.mark.skipif(not reference_available(), reason="needs the reference archive (oracle/_ref) or /root/reference")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _model(kw, seed, precision):
    m = Model(**kw, precision=precision)
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


def _grads(m, fwd, x, t, seed=11, **kw):
    for p in m.parameters():
        p.grad = None
    x = x.clone().requires_grad_(True)
    y = fwd(x, t, **kw)
    w = make_input("gw", tuple(y.shape), seed=seed).to(y.device)
    (y * w).sum().backward()
    torch.cuda.synchronize()
    return y.detach(), x.grad.clone(), {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}


@needs_ref
@pytest.mark.parametrize("tag,kw,b,n,n_p,n_c", [
    ("cond_d512_L12_b2_n512", dict(dim=512, depth=12, dim_prompt=512, condition_on_prompt=True, cond_drop_prob=0.), 2, 512, 103, 512),
    ("cond_d128_L2_b3_n300_proj", dict(dim=128, depth=2, dim_prompt=96, condition_on_prompt=True, cond_drop_prob=0.), 3, 300, 53, 280)])
def test_every_gradient_of_the_conditioned_model_matches_the_reference_autograd(tag, kw, b, n, n_p, n_c):
    """BASELINE config 3's architecture under autograd (NS2:1635, 1886): `.grad` of EVERY parameter -- the perceiver resampler's, the
    cross attentions', `null_*` -- and of x through the HIP training path against the unmodified upstream `Model` on the same GPU"""
    ns2 = load_reference()
    m = Model(**kw, precision="hybrid")
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=81)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    ref = ns2.Model(**kw)
    ref.load_state_dict(sd)
    ref = ref.to(DEV).train()
    x = make_input("x", (b, n, kw["dim"]), seed=82).to(DEV)
    t = make_input("times", (b,), seed=82, uniform=True).to(DEV)
    prompt = make_input("prompt", (b, n_p, kw["dim_prompt"]), seed=83).to(DEV)
    cond = make_input("cond", (b, kw["dim_prompt"], n_c), seed=83).to(DEV)
    y1, dx1, g1 = _grads(m, m, x, t, prompt=prompt, cond=cond)
    y0, dx0, g0 = _grads(ref, ref, x, t, prompt=prompt, cond=cond)
    assert set(g1) == set(g0)
    errs = {}
    for k, g in g0.items():
        if g is None:                                   # null_* parameters: selected by torch.where with a false mask
            assert g1[k] is None or float(g1[k].abs().max()) == 0.0, k
            continue
        assert g1[k] is not None, k
        errs[k] = rel(g1[k], g) if float(g.abs().max()) > 0 else float(g1[k].abs().max())
    worst = max(errs.items(), key=lambda kv: kv[1])
    rec = dict(n_tensors=len(errs), worst_param=worst[0], worst_rel=worst[1], x_grad_rel=rel(dx1, dx0), out_rel=rel(y1, y0),
               median_rel=sorted(errs.values())[len(errs) // 2],
               worst_resampler=max((e for k, e in errs.items() if "perceiver_resampler" in k), default=0.0))
    record(f"backward_vs_reference_autograd/{tag}", rec)
    print(tag, rec)
    assert rec["out_rel"] < 1e-4 and rec["x_grad_rel"] < 1e-3, rec
    bad = {k: e for k, e in errs.items() if not e < 1e-3}
    assert not bad, bad


def test_frozen_model_returns_dx_without_weight_gradients():
    kw = dict(dim=128, depth=2)
    m, _ = _model(kw, 91, "hybrid")
    m = m.train()
    x = make_input("x", (2, 256, 128), seed=92).to(DEV)
    t = make_input("times", (2,), seed=92, uniform=True).to(DEV)
    _, dx_all, g_all = _grads(m, m, x, t)
    for p in m.parameters():
        p.requires_grad_(False)
    _, dx_frozen, g_frozen = _grads(m, m, x, t)
    assert all(g is None for g in g_frozen.values())
    assert torch.equal(dx_all, dx_frozen)               # the same kernels in the same order produce dL/dx
